// "Expand + join" kernel: the last 1x1 convolution of a bottleneck (resnet.py:123, Co = 4 * Ci) with the block's closing
// BatchNorm, residual join and ReLU (resnet.py:125-133) in its epilogue, for the case where the BatchNorm constants are known
// before the convolution runs (vince_bn_gram_finalize):
//
//     out[p][co] = relu( scale[co] * sum_k W[co][k] x[p][k] + shift[co] + identity'[p][co] )
//     identity'  = id_scale ? identity * id_scale[co] + id_shift[co] : identity          (out may alias identity)
//
// This is a pure HBM stream -- K = 64 / 128, so 2 * K FLOP per output element against >= 4 bytes moved -- and is built as one:
//   * persistent workgroups, one per CU, each owning a fixed group of 256 output channels whose weights stay in LDS
//     (32 KB at K = 64, 64 KB at K = 128) for the lifetime of the launch;
//   * a dedicated LOADER wavefront streams the 128-pixel x K input tiles HBM -> LDS by LDS-DMA into a ring (its vmcnt holds
//     nothing but those loads, so counted waits are exact), ahead of the eight CONSUMER wavefronts;
//   * each consumer owns 64 pixels x 64 channels as 2 x 2 MFMA 32x32x16 tiles; the accumulator layout (lane = pixel, register
//     quad = 4 consecutive channels, lane halves interleaved) is turned into whole 16-byte channel chunks per lane with
//     v_permlane32_swap, so identity loads and output stores go straight between registers and HBM -- no LDS transposition, no
//     barrier in the epilogue, and (the statistics come from the Gram matrix) no reduction;
//   * the identity chunks of a tile are requested before its MFMA loop and the stores of tile t drain while tile t + 1 computes.
#include <string.h>

#include "common.h"

namespace {

constexpr int XJ_PX = 128;          // pixels per tile (2 consumer rows of 64)
constexpr int XJ_CG = 256;          // channels per workgroup (4 consumer columns of 64)
constexpr int XJ_CONSUMERS = 8;
constexpr int XJ_THREADS = (XJ_CONSUMERS + 1) * 64;

struct XjParams {
    const void* x;
    const void* w;
    const void* identity;
    void* out;
    void* y_raw;          // optional: the convolution's own output, rounded to bf16 (what BatchNorm backward reads)
    uint8_t* mask_out;    // optional: one byte per 16-byte chunk of out, bit e = pre-ReLU value e > 0 (as vince_bn_apply writes)
    const float* out_scale;
    const float* out_shift;
    const float* id_scale;
    const float* id_shift;
    double* stats;        // PLAIN mode: double[replicas][Co][2] per-channel (sum, sum of squares) of the stored output
    // DGRAD mode (the block-input gradient: out = conv + (acc_mask bit ? out : 0), plus the BatchNorm-backward sums of the
    // BatchNorm that consumes `out`): see vince_conv_expand_dgrad
    const uint8_t* acc_mask;
    const uint8_t* out_mask;   // DGRAD: the stored value is gated by these bits (the ReLU of the block below); br_sums without br_y then
                               // receives the plain per-channel (sum, sum of squares) of the stored values (vince_conv_expand_dgrad_masked)
    const void* br_y;
    const uint8_t* br_bits;
    const float* br_mean;
    const float* br_invstd;
    double* br_sums;
    // NEXT mode (the join + the FOLLOWING bottleneck's conv1, vince_conv_expand_join_next): y2[p][c1] = sum_k w2[c1][k] out[p][k]
    const void* w2;       // [64][256] bf16
    void* y2;             // [rows][64] bf16: the next conv1's raw output
    double* stats2;       // double[replicas2][64][2]: (sum, sum of squares) of the stored y2
    const float* bias2;   // optional (folded inference): y2 = [relu]( bf16(conv) + bias2[c] ), as vince_conv_igemm's bias epilogue rounds
    int relu2;
    uint32_t w2_bytes;
    int replicas2;
    uint32_t rows, Co, x_bytes, w_bytes;
    int ptiles, cgroups, relu, replicas;
};

constexpr int XJ_K2 = 256;          // NEXT mode: reduction length of the fused second convolution = this launch's Co

// NEXT = output channels of the fused second convolution: 0 (none), 64 (layer1's conv1: 256 -> 64) or 128 (layer2.0's conv1: 256 -> 128)
template <int K, int STAGES, int NEXT = 0>
struct XjSmem {
    static constexpr int NKT = K / 32;                      // 64-byte K blocks per row
    static constexpr int WB = NKT * XJ_CG * 64;             // resident weights
    static constexpr int XB = NKT * XJ_PX * 64;             // one input tile
    // scale, shift, id_scale, id_shift of this channel group.  NEXT = 128 has no room for it (32 KB conv3 weights + 2 x 16 KB ring + 32 KB
    // stripes + 64 KB of next-conv weights = the 160 KB of a CU exactly): that instantiation reads scale / shift from memory.
    static constexpr int TAB = NEXT == 128 ? 0 : 4 * XJ_CG * 4;
    // per-consumer transposition buffer for the stores: 32 pixels x CHW channels.  A 16-byte store per lane with lane = pixel
    // reaches memory as 32-byte pieces of 32 different rows, and partial-line WRITES are what this chip's L2 charges for
    // (measured: ~18 us per million 32-byte write requests, reads nearly free); through the buffer 8 (4) consecutive lanes write
    // one whole 128-byte line (64-byte half line where LDS is short: K = 128 keeps 64 KB of weights and a 64 KB ring).
    static constexpr int CHW = K == 64 ? 64 : 32;
    static constexpr int TBUF = 32 * CHW * 2;
    static constexpr int OFF_T = WB + STAGES * XB + TAB;
    // NEXT: the eight per-consumer buffers become two SHARED ones (one per 64-pixel half): 32 pixels x all 256 output channels, 512-byte
    // rows, 16-byte chunk c of row r at slot c ^ (r & 15) -- the four consumers of a half write their 64-channel stripes, store them
    // from there as before, and two of them then read whole rows as the MFMA operand of the second convolution.  Same 32 KB.
    static constexpr int ABUF = 32 * XJ_K2 * 2;
    static constexpr int OFF_W2 = OFF_T + XJ_CONSUMERS * TBUF;
    static constexpr int BYTES = OFF_W2 + NEXT * XJ_K2 * 2;
    static_assert(!NEXT || (K == 64 && 2 * ABUF == XJ_CONSUMERS * TBUF), "NEXT: K = 64 only");
    static_assert(BYTES <= 163840, "LDS");
};

// PLAIN: the same streaming structure for an expand convolution on its own (resnet.py:123 without the join: grad-enabled
// forwards, the stride-1 downsample conv of layer1): out = the raw convolution output, no identity, no constants; the BatchNorm
// statistics of the stored values are kept per lane -- after the transposition a lane owns the same 8 channels for the whole
// launch -- and leave as one fp64 atomic per channel per wavefront at the end (the implicit-GEMM epilogue pays a shuffle + LDS
// reduction and 2 x 128 atomics per 128-pixel tile: 24 % of the layer1 expand conv).
// DGRAD: the input gradient of a bottleneck's conv1 (the "expand" shape again: dx[p][4w] from dy[p][w]) with the epilogues of
// vince_conv_igemm's gradient instantiation -- the residual-gradient join  out = dgrad + (acc_mask bit ? out_old : 0)  in place,
// and the (sum g', sum g' xhat) reduction of the BatchNorm below that consumes `out` (g' = out gated by ITS ReLU bits, xhat from
// its saved conv output) -- the join in the MFMA layout with out_old requested a unit ahead, the reduction after the
// transposition where a lane owns the same 8 channels for the whole launch.
// SAVE: 0 nothing beside `out`; 1 the ReLU mask bytes (the forward of the BatchNorm-backward algebra); 2 mask bytes + the raw conv output
// NEXT: the FOLLOWING bottleneck's first convolution (resnet.py:117, 1x1, 256 -> 64, layer1) on the block output while it is in LDS:
// its 411 MB re-read from HBM -- the whole cost of that launch -- goes.  Per 32-pixel unit: the four consumers of a pixel half write
// their stripes of `out` into the shared buffer (barrier), two of them multiply the 32 x 256 rows with the resident 64 x 256 weights
// (one 32-channel tile each, 16 x v_mfma_f32_32x32x16_bf16, K ascending: the same products in the same order as vince_conv_igemm's
// launch, so y2 is BIT-IDENTICAL to it), store y2 as 64-byte row pieces straight from the accumulators (lane = channel) and keep
// that channel's (sum, sum of squares) in two registers for the whole launch.  Four barriers per tile instead of one (the loader
// counts along): stripes of unit 0 complete / read / stripes of unit 1 complete / (next tile's) read.
// NEXT = 128 (the layer1 -> layer2 transition, 256 -> 128): all four consumers of a half multiply (one 32-channel tile each); two ring stages.
template <int K, int STAGES, bool ID_AFFINE, int SAVE, bool PLAIN = false, bool DGRAD = false, int NEXT = 0>
__global__ __launch_bounds__(XJ_THREADS) void conv_xjoin_kernel(const XjParams p) {
    using S = XjSmem<K, STAGES, NEXT>;
    static_assert(!NEXT || (!PLAIN && !DGRAD), "NEXT rides on the join");
    static_assert(NEXT != 128 || !ID_AFFINE, "NEXT = 128 keeps no constant table");
    constexpr bool TABLE = S::TAB != 0;
    constexpr int NKT = S::NKT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::BYTES];
    unsigned char* const wsm = smem;
    unsigned char* const xsm = smem + S::WB;
    float* const tab = (float*)(smem + S::WB + STAGES * S::XB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (channel group, pixel-tile lane), fixed for its lifetime (grid % cgroups == 0).  The hardware deals consecutive
    // workgroups to the 8 XCDs round-robin: the channel groups of ONE pixel-tile lane are put on the same XCD, so that the input
    // tile comes from HBM once and from that XCD's L2 for the other groups (b % cgroups would put them on different XCDs).
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
    const int c0 = cg * XJ_CG;
    const int ntiles = first < p.ptiles ? (p.ptiles - first + step - 1) / step : 0;   // pixel tiles first, first + step, ...

    const v4i_t rsrc_x = make_rsrc(p.x, p.x_bytes);
    const v4i_t rsrc_w = make_rsrc(p.w, p.w_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr uint32_t OOB = 0x80000000u;

    // lane -> (row dr of a 16-row x 64-byte DMA piece, slot); the slot holds logical K chunk slot ^ ((row >> 2) & 3), the
    // XOR swizzle that keeps the ds_read_b128 fragment reads of 32 consecutive rows off each other's banks
    const int dr = lane >> 2, dslot = lane & 3;
    const int dchunk = dslot ^ ((dr >> 2) & 3);

    // (tab: id_scale / id_shift rows only matter to the ID_AFFINE instantiation)
    // ---- resident weights + constant table (every wavefront helps, once) -------------------------------------------------
    {
        constexpr int PIECES = NKT * XJ_CG / 16;            // 1 KB pieces
        for (int pc = wave; pc < PIECES; pc += XJ_CONSUMERS + 1) {
            const int kt = pc / (XJ_CG / 16), rb = (pc % (XJ_CG / 16)) * 16;
            const uint32_t co = (uint32_t)(c0 + rb + dr);
            const uint32_t off = co < p.Co ? (co * (uint32_t)K + (uint32_t)(kt * 32 + dchunk * 8)) * 2u : OOB;
            lds_dma16(__builtin_amdgcn_readfirstlane(smem_base + kt * (XJ_CG * 64) + rb * 64), off, rsrc_w);
        }
        if constexpr (NEXT) {
            // w2 rows of 512 bytes; one DMA instruction = 2 rows; lane -> (row, physical slot), logical chunk = slot ^ (row & 15)
            const v4i_t rsrc_w2 = make_rsrc(p.w2, p.w2_bytes);
            for (int pc = wave; pc < NEXT / 2; pc += XJ_CONSUMERS + 1) {
                const int r = pc * 2 + (lane >> 5), slot = lane & 31;
                const uint32_t off = (uint32_t)(r * XJ_K2 + ((slot ^ (r & 15)) * 8)) * 2u;
                lds_dma16(__builtin_amdgcn_readfirstlane(smem_base + S::OFF_W2 + pc * 1024), off, rsrc_w2);
            }
        }
        for (int i = tid; DGRAD && i < XJ_CG; i += XJ_THREADS) {
            const int c = c0 + i;
            const bool ok = (uint32_t)c < p.Co && p.br_y != nullptr;
            tab[i] = ok ? p.br_mean[c] : 0.f;
            tab[XJ_CG + i] = ok ? p.br_invstd[c] : 0.f;
        }
        for (int i = tid; TABLE && !PLAIN && !DGRAD && i < XJ_CG; i += XJ_THREADS) {
            const int c = c0 + i;
            const bool ok = (uint32_t)c < p.Co;
            tab[i] = ok ? p.out_scale[c] : 0.f;
            tab[XJ_CG + i] = ok ? p.out_shift[c] : 0.f;
            tab[2 * XJ_CG + i] = (ok && p.id_scale) ? p.id_scale[c] : 1.f;
            tab[3 * XJ_CG + i] = (ok && p.id_scale) ? p.id_shift[c] : 0.f;
        }
        wait_vmcnt<0>();
        __syncthreads();
    }

    constexpr int X_PIECES = NKT * XJ_PX / 16;              // DMA instructions per input tile (16 at K = 64, 32 at K = 128)
    if (wave == XJ_CONSUMERS) {
        // =============================== loader ===============================
        // invariant at barrier B(t): tile t has landed; tiles t+1 .. t+STAGES-2 are in flight; the ring stage of tile t-1 is free
        // once the barrier releases (every consumer finished tile t-1's MFMA loop before arriving).
        auto issue_x = [&](int t) {
            if (t >= ntiles) return;
            const uint32_t p0 = (uint32_t)(first + t * step) * XJ_PX;
            const uint32_t sbase = smem_base + S::WB + (uint32_t)(t % STAGES) * S::XB;
#pragma unroll
            for (int pc = 0; pc < X_PIECES; ++pc) {
                const int kt = pc / (XJ_PX / 16), rb = (pc % (XJ_PX / 16)) * 16;
                const uint32_t m = p0 + rb + dr;
                const uint32_t off = m < p.rows ? (m * (uint32_t)K + (uint32_t)(kt * 32 + dchunk * 8)) * 2u : OOB;
                lds_dma16(__builtin_amdgcn_readfirstlane(sbase + kt * (XJ_PX * 64) + rb * 64), off, rsrc_x);
            }
        };
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t) issue_x(t);
        for (int t = 0; t < ntiles; ++t) {
            // with 3 stages tile t+1 (X_PIECES instructions, issued after tile t's) may stay in flight across the barrier
            if (STAGES == 3 && t + 1 < ntiles) wait_vmcnt<X_PIECES>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                   // B(t)
            issue_x(t + STAGES - 1);                        // into the stage tile t-1 occupied
            if constexpr (NEXT) {                           // the consumers' three stripe barriers of this tile
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
        }
        return;
    }

    // =============================== consumers ===============================
    // Work unit = one 32-pixel column block of the wavefront's 64 x 64 tile (i = 0, 1): 32 accumulator registers and four
    // 16-byte identity chunks.  The identity chunks of unit u + 1 are requested BEFORE unit u's epilogue issues its stores:
    // vector memory operations retire in order, so waiting for those loads later never waits for the younger stores, which
    // drain in the background while the next unit computes.
    const int wp = wave >> 2, wc = wave & 3;                // 64-pixel half, 64-channel quarter
    const int sw = ((lane & 31) >> 2) & 3, khalf = lane >> 5;
    const int row_off = (lane & 31) * 64;
    const bf16_t* __restrict__ idn = (const bf16_t*)p.identity;
    bf16_t* __restrict__ out = (bf16_t*)p.out;
    bf16_t* __restrict__ yraw = (bf16_t*)p.y_raw;
    const unsigned char* const wfrag = wsm + (wc * 64) * 64 + row_off;
    const uint32_t lane_px = (uint32_t)(wp * 64 + (lane & 31));
    const size_t ch_off = (size_t)(c0 + wc * 64 + khalf * 8);

    // lane's pixel of unit (t, i) and the element offset of its first chunk; chunks (j, gp) follow at + 32 j + 16 gp
    auto unit_off = [&](int t, int i, bool& ok) -> size_t {
        const uint32_t pix = (uint32_t)(first + t * step) * XJ_PX + lane_px + 32u * (uint32_t)i;
        ok = t < ntiles && pix < p.rows;
        return (size_t)pix * p.Co + ch_off;
    };
    auto load_ids = [&](uint4 (&dst)[2][2], int t, int i) {
        if constexpr (PLAIN) return;
        bool ok;
        const size_t off = unit_off(t, i, ok);
        if constexpr (DGRAD) ok = ok && p.relu;            // (p.relu doubles as "accumulate" in DGRAD mode)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) dst[j][gp] = ok ? *(const uint4*)(idn + off + j * 32 + gp * 16) : make_uint4(0, 0, 0, 0);
    };
    unsigned char* const tbuf = smem + S::OFF_T + (NEXT ? wp * S::ABUF : wave * S::TBUF);
    constexpr int CHW = S::CHW, CPR = CHW / 8;              // channels / 16-byte chunks per buffer row
    constexpr int NPASS = 64 / CHW;                         // buffer passes per unit (1: whole 64-channel rows, 2: per MFMA tile j)
    constexpr int LPR = CPR;                                // lanes that share a pixel row when storing
    constexpr int NST = 32 / (64 / LPR);                    // store instructions per pass
    // one pass: chunks pk[jj][gp] (lane = pixel) -> buffer -> (lane group = pixel row) -> global; optional mask bytes
    float ssum[NPASS][8], ssq[NPASS][8];                   // PLAIN: statistics of this lane's channels (chunk lane % LPR of each pass)
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) ssum[q][e] = ssq[q][e] = 0.f;
    uint4 bry[NPASS][NST];                                  // DGRAD: the consumer BatchNorm's saved conv output at this lane's store
    uint32_t brb[NPASS][NST];                               // positions (requested at the top of the unit) and its ReLU bits
    auto stage_store = [&](const uint4 (&pk)[2][2], int pass, bf16_t* __restrict__ dst, uint8_t* __restrict__ mdst, uint32_t pix0) {
        const int row = lane & 31;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jj = 0; jj < 2 / NPASS; ++jj)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int j = NPASS == 1 ? jj : pass;
                const int cpos = (NPASS == 1 ? j * 4 : 0) + 2 * gp + khalf;
                const int swz = NPASS == 1 ? (row & 7) : ((row >> 1) & 3);
                if constexpr (NEXT) *(uint4*)(tbuf + row * (XJ_K2 * 2) + (((wc * 8 + cpos) ^ (row & 15)) * 16)) = pk[j][gp];
                else *(uint4*)(tbuf + row * (CHW * 2) + ((cpos ^ swz) * 16)) = pk[j][gp];
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int sidx = 0; sidx < NST; ++sidx) {
            const int prow = lane / LPR + (64 / LPR) * sidx, c = lane % LPR;
            const int swz = NPASS == 1 ? (prow & 7) : ((prow >> 1) & 3);
            const uint4 val = NEXT ? *(const uint4*)(tbuf + prow * (XJ_K2 * 2) + (((wc * 8 + c) ^ (prow & 15)) * 16))
                                   : *(const uint4*)(tbuf + prow * (CHW * 2) + ((c ^ swz) * 16));
            const uint32_t pix = pix0 + (uint32_t)prow;
            if (pix < p.rows) {
                const size_t off = (size_t)pix * p.Co + (size_t)(c0 + wc * 64 + (NPASS == 1 ? 0 : pass * 32) + c * 8);
                *(uint4*)(dst + off) = val;
                if constexpr (PLAIN) {
                    float f[8];
                    Chunk<bf16_t>::unpack(val, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ssum[pass][e] += f[e]; ssq[pass][e] += f[e] * f[e]; }
                }
                if constexpr (DGRAD) {
                    if (!p.br_y && p.br_sums) {             // (uniform) plain column sums of the stored (already gated) gradient
                        float g[8];
                        Chunk<bf16_t>::unpack(val, g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ssum[pass][e] += g[e]; ssq[pass][e] += g[e] * g[e]; }
                    }
                    if (p.br_y) {                           // (uniform) sums of the STORED gradient, as vince_bn_bwd_reduce defines them
                        float g[8], yy[8];
                        Chunk<bf16_t>::unpack(val, g);
                        Chunk<bf16_t>::unpack(bry[pass][sidx], yy);
                        int tb = wc * 64 + (NPASS == 1 ? 0 : pass * 32) + c * 8;
                        asm volatile("" : "+v"(tb));
                        const float4 m0 = *(const float4*)(tab + tb), m1 = *(const float4*)(tab + tb + 4);
                        const float4 i0 = *(const float4*)(tab + XJ_CG + tb), i1 = *(const float4*)(tab + XJ_CG + tb + 4);
                        const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                        const float is[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
                        const uint32_t bb = brb[pass][sidx];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float ge = ((bb >> e) & 1u) ? g[e] : 0.f;
                            ssum[pass][e] += ge;
                            ssq[pass][e] += ge * (yy[e] - mu[e]) * is[e];
                        }
                    }
                }
                if (mdst) {                                 // bit e = stored value e > 0 (== pre-ReLU value > 0)
                    const uint32_t wv[4] = {val.x, val.y, val.z, val.w};
                    uint32_t bits = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t lo16 = wv[e] & 0xffffu, hi16 = wv[e] >> 16;
                        bits |= ((lo16 != 0 && !(lo16 & 0x8000u)) ? 1u : 0u) << (2 * e);
                        bits |= ((hi16 != 0 && !(hi16 & 0x8000u)) ? 1u : 0u) << (2 * e + 1);
                    }
                    mdst[off / 8] = (uint8_t)bits;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    auto run_unit = [&](const uint4 (&ids)[2][2], int t, int i) {
        uint2 amask = make_uint2(0xffffffffu, 0xffffffffu);  // DGRAD: the 8 acc_mask bytes of this lane's pixel (64 channels)
        uint2 omask = make_uint2(0xffffffffu, 0xffffffffu);  // DGRAD: the 8 out_mask bytes likewise
        if constexpr (DGRAD) {
            const uint32_t pixu = (uint32_t)(first + t * step) * XJ_PX + (uint32_t)(wp * 64 + i * 32);
            if (p.acc_mask) {
                const uint32_t pix = pixu + (uint32_t)(lane & 31);
                if (t < ntiles && pix < p.rows) amask = *(const uint2*)(p.acc_mask + ((size_t)pix * p.Co + (size_t)(c0 + wc * 64)) / 8);
            }
            if (p.out_mask) {
                const uint32_t pix = pixu + (uint32_t)(lane & 31);
                if (t < ntiles && pix < p.rows) omask = *(const uint2*)(p.out_mask + ((size_t)pix * p.Co + (size_t)(c0 + wc * 64)) / 8);
            }
            if (p.br_y) {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
                    for (int sidx = 0; sidx < NST; ++sidx) {
                        const int prow = lane / LPR + (64 / LPR) * sidx, c = lane % LPR;
                        const uint32_t pix = pixu + (uint32_t)prow;
                        const size_t off = (size_t)pix * p.Co + (size_t)(c0 + wc * 64 + (NPASS == 1 ? 0 : pass * 32) + c * 8);
                        const bool ok = t < ntiles && pix < p.rows;
                        bry[pass][sidx] = ok ? *(const uint4*)((const bf16_t*)p.br_y + off) : make_uint4(0, 0, 0, 0);
                        brb[pass][sidx] = (ok && p.br_bits) ? (uint32_t)p.br_bits[off / 8] : 0xffu;
                    }
            }
        }
        const unsigned char* xfrag = xsm + (t % STAGES) * S::XB + (wp * 64 + i * 32) * 64 + row_off;
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int slot = ((s2 * 2 + khalf) ^ sw) * 16;
                const uint4 xf = *(const uint4*)(xfrag + kt * (XJ_PX * 64) + slot);
                bf16x8_t bv;
                __builtin_memcpy(&bv, &xf, 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint4 wf = *(const uint4*)(wfrag + kt * (XJ_CG * 64) + j * 32 * 64 + slot);
                    bf16x8_t av;
                    __builtin_memcpy(&av, &wf, 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
        uint4 opk[2][2], rpk[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // quads g0 = 2 gp (registers 8 gp .. +3) and g1 = 2 gp + 1 (8 gp + 4 .. +7): after the swap the low half-wave
                // holds chunk g0 whole (its own 4 channels + the high half's 4), the high half-wave chunk g1 = 2 gp + khalf
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][8 * gp + e]),
                                                                    __float_as_uint(acc[j][8 * gp + 4 + e]), false, false);
                    v[e] = __uint_as_float(r[0]);
                    v[4 + e] = __uint_as_float(r[1]);
                }
                if constexpr (SAVE == 2) rpk[j][gp] = Chunk<bf16_t>::pack(v);
                if constexpr (PLAIN) {
                    opk[j][gp] = Chunk<bf16_t>::pack(v);
                    continue;
                }
                if constexpr (DGRAD) {
                    float o[8];
                    Chunk<bf16_t>::unpack(ids[j][gp], o);
                    const int bidx = 4 * j + 2 * gp + khalf;                     // this chunk's byte among the pixel's 8
                    const uint32_t ab = ((bidx < 4 ? amask.x : amask.y) >> (8 * (bidx & 3))) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += ((ab >> e) & 1u) ? o[e] : 0.f;
                    const uint32_t ob = ((bidx < 4 ? omask.x : omask.y) >> (8 * (bidx & 3))) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((ob >> e) & 1u) ? v[e] : 0.f;
                    opk[j][gp] = Chunk<bf16_t>::pack(v);
                    continue;
                }
                int tb = wc * 64 + j * 32 + (2 * gp + khalf) * 8;
                asm volatile("" : "+v"(tb));                // keep the table reads here: hoisted out of the tile loop they would
                                                            // pin 64-128 registers for the lifetime of the wavefront
                float idf[8];
                Chunk<bf16_t>::unpack(ids[j][gp], idf);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 a, b;
                    if constexpr (TABLE) {
                        a = *(const float4*)(tab + tb + 4 * h);
                        b = *(const float4*)(tab + XJ_CG + tb + 4 * h);
                    } else {                                // (one channel group, Co = 256: every channel of the table exists)
                        a = *(const float4*)(p.out_scale + c0 + tb + 4 * h);
                        b = *(const float4*)(p.out_shift + c0 + tb + 4 * h);
                    }
                    float4 c = make_float4(1.f, 1.f, 1.f, 1.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (ID_AFFINE) {
                        c = *(const float4*)(tab + 2 * XJ_CG + tb + 4 * h);
                        d = *(const float4*)(tab + 3 * XJ_CG + tb + 4 * h);
                    }
                    v[4 * h + 0] = v[4 * h + 0] * a.x + b.x + (ID_AFFINE ? idf[4 * h + 0] * c.x + d.x : idf[4 * h + 0]);
                    v[4 * h + 1] = v[4 * h + 1] * a.y + b.y + (ID_AFFINE ? idf[4 * h + 1] * c.y + d.y : idf[4 * h + 1]);
                    v[4 * h + 2] = v[4 * h + 2] * a.z + b.z + (ID_AFFINE ? idf[4 * h + 2] * c.z + d.z : idf[4 * h + 2]);
                    v[4 * h + 3] = v[4 * h + 3] * a.w + b.w + (ID_AFFINE ? idf[4 * h + 3] * c.w + d.w : idf[4 * h + 3]);
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                opk[j][gp] = Chunk<bf16_t>::pack(v);
            }
        if (t < ntiles) {                                   // (uniform)
            const uint32_t pix0 = (uint32_t)(first + t * step) * XJ_PX + (uint32_t)(wp * 64 + i * 32);
            if constexpr (NEXT) {
                // the shared buffer still holds unit 0's rows for the two wavefronts multiplying them: nobody overwrites before they are done
                if (i == 1) __builtin_amdgcn_s_barrier();
                if constexpr (SAVE == 2) stage_store(rpk, 0, yraw, nullptr, pix0);   // raw output first: the buffer must END UP holding `out`
                stage_store(opk, 0, out, SAVE ? p.mask_out : nullptr, pix0);
            } else {
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    stage_store(opk, pass, out, SAVE ? p.mask_out : nullptr, pix0);
                    if constexpr (SAVE == 2) stage_store(rpk, pass, yraw, nullptr, pix0);
                }
            }
        }
    };
    // NEXT: the second convolution of unit (t, i) for this pixel half, channel tile c1t -- NEXT = 64: tile (wc & 1), run by the consumers
    // with (wc >> 1) == i; NEXT = 128: tile wc, every consumer
    const int c1t = NEXT == 128 ? wc : (wc & 1);
    float s2 = 0.f, q2 = 0.f;                               // statistics of channel (wc & 1) * 32 + (lane & 31), this lane's pixel rows
    auto gemm2 = [&](int t, int i) {
        const int sz = lane & 15;                           // (row & 15) of both operands' rows: lane & 31 within a 32-row tile
        const unsigned char* const arow = tbuf + (lane & 31) * (XJ_K2 * 2);
        const unsigned char* const brow = smem + S::OFF_W2 + (c1t * 32 + (lane & 31)) * (XJ_K2 * 2);
        f32x16_t c2;
#pragma unroll
        for (int e = 0; e < 16; ++e) c2[e] = 0.f;
#pragma unroll 4
        for (int s = 0; s < XJ_K2 / 16; ++s) {
            const int slot = ((2 * s + khalf) ^ sz) * 16;
            const uint4 af = *(const uint4*)(arow + slot), bf = *(const uint4*)(brow + slot);
            bf16x8_t av, bv;
            __builtin_memcpy(&av, &af, 16);
            __builtin_memcpy(&bv, &bf, 16);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
        }
        // C[pixel][channel]: lane = channel column, register r = pixel row (r & 3) + 8 (r >> 2) + 4 khalf: 32 lanes store 64 consecutive bytes
        const uint32_t pix0 = (uint32_t)(first + t * step) * XJ_PX + (uint32_t)(wp * 64 + i * 32);
        bf16_t* __restrict__ y2 = (bf16_t*)p.y2 + (size_t)(c1t * 32 + (lane & 31));
        if (p.bias2) {                                      // (uniform) the implicit-GEMM epilogue's order: round, add the bias, ReLU, round
            const float b2 = p.bias2[c1t * 32 + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t u0 = pack_bf16x2(c2[r], c2[r + 1]);
                float f0 = __uint_as_float(u0 << 16) + b2, f1 = __uint_as_float(u0 & 0xffff0000u) + b2;
                if (p.relu2) { f0 = fmaxf(f0, 0.f); f1 = fmaxf(f1, 0.f); }
                c2[r] = f0;
                c2[r + 1] = f1;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const uint32_t u = pack_bf16x2(c2[r], c2[r + 1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t pix = pix0 + (uint32_t)(((r + h) & 3) + 8 * ((r + h) >> 2) + 4 * khalf);
                if (pix < p.rows) {
                    const uint32_t b16 = h ? (u >> 16) : (u & 0xffffu);
                    y2[(size_t)pix * NEXT] = (bf16_t)b16;
                    const float f = __uint_as_float(b16 << 16);
                    s2 += f;
                    q2 += f * f;
                }
            }
        }
    };

    {
        // identity / old-gradient chunks one UNIT ahead of the unit being computed (four 16-byte chunks per lane in flight).  A whole tile
        // ahead (round 2) held 64 registers for it and pushed every join instantiation over the 168 the nine-wavefront workgroup allows:
        // 12-32 bytes of scratch per lane, and a scratch RELOAD waits for every older vector memory operation of the wavefront -- the
        // output stores it has just issued -- once per unit.
        uint4 oA[2][2], oB[2][2];
        load_ids(oA, 0, 0);
        for (int t = 0; t < ntiles; ++t) {
            load_ids(oB, t, 1);
            __builtin_amdgcn_s_barrier();                   // B(t): the loader has seen tile t land
            run_unit(oA, t, 0);
            load_ids(oA, t + 1, 0);
            if constexpr (NEXT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();               // every stripe of unit 0 is in the shared buffers
                if (NEXT == 128 || (wc >> 1) == 0) gemm2(t, 0);
            }
            run_unit(oB, t, 1);
            if constexpr (NEXT) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();               // ... of unit 1 (B(t + 1) keeps the next tile's stripes behind these reads)
                if (NEXT == 128 || (wc >> 1) == 1) gemm2(t, 1);
            }
        }
    }
    if constexpr (NEXT) {
        if (p.stats2) {                                     // the two lane halves hold the same channel: fold, one fp64 atomic pair per channel and wavefront
            s2 += __shfl_xor(s2, 32, 64);
            q2 += __shfl_xor(q2, 32, 64);
            if (lane < 32) {
                double* dst = p.stats2 + ((size_t)(blockIdx.x % (unsigned)p.replicas2) * NEXT + (size_t)(c1t * 32 + lane)) * 2;
                unsafeAtomicAdd(dst, (double)s2);
                unsafeAtomicAdd(dst + 1, (double)q2);
            }
        }
    }
    if constexpr (PLAIN || DGRAD) {
        if (DGRAD ? (p.br_sums != nullptr) : (p.stats != nullptr)) {
            // lanes l, l + LPR, l + 2 LPR, ... hold the same channels: fold them, then one fp64 atomic per channel and wavefront
#pragma unroll
            for (int q = 0; q < NPASS; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int o = LPR; o < 64; o <<= 1) {
                        ssum[q][e] += __shfl_xor(ssum[q][e], o, 64);
                        ssq[q][e] += __shfl_xor(ssq[q][e], o, 64);
                    }
            if (lane < LPR) {
                double* dst = (DGRAD ? p.br_sums : p.stats) + (size_t)(blockIdx.x % (unsigned)p.replicas) * p.Co * 2;
#pragma unroll
                for (int q = 0; q < NPASS; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ch = c0 + wc * 64 + (NPASS == 1 ? 0 : q * 32) + lane * 8 + e;
                        unsafeAtomicAdd(dst + (size_t)ch * 2, (double)ssum[q][e]);
                        unsafeAtomicAdd(dst + (size_t)ch * 2 + 1, (double)ssq[q][e]);
                    }
            }
        }
    }
}

}  // namespace

// CU count of the current device, asked once (the property query is a host call worth ~100 us)
static int xj_num_cu() {
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

extern "C" int vince_conv_expand_stats(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co, void* out,
                                       double* stats, int32_t replicas, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv_expand_stats: bf16 only");
    VINCE_CHECK_ARG(x && w && out && rows > 0, VINCE_E_ARG, "vince_conv_expand_stats: null pointer");
    VINCE_CHECK_ARG(K == 64 || K == 128, VINCE_E_UNSUPPORTED, "vince_conv_expand_stats: K=%d (64 or 128)", K);
    VINCE_CHECK_ARG(Co > 0 && Co % XJ_CG == 0, VINCE_E_SHAPE, "vince_conv_expand_stats: Co=%d must be a multiple of %d", Co, XJ_CG);
    VINCE_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN, "vince_conv_expand_stats: pointers must be 16-byte aligned");
    const unsigned long long xb = (unsigned long long)rows * K * 2, wb = (unsigned long long)Co * K * 2;
    VINCE_CHECK_ARG(xb < 0x7ff00000ull && rows < (1ll << 31), VINCE_E_UNSUPPORTED, "vince_conv_expand_stats: input beyond the 31-bit buffer offsets");
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    XjParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.out = out; p.stats = stats; p.replicas = replicas;
    p.rows = (uint32_t)rows; p.Co = (uint32_t)Co; p.x_bytes = (uint32_t)xb; p.w_bytes = (uint32_t)wb;
    p.ptiles = (int)((rows + XJ_PX - 1) / XJ_PX);
    p.cgroups = Co / XJ_CG;
    long grid = xj_num_cu();
    const long items = (long)p.ptiles * p.cgroups;
    if (grid > items) grid = items;
    grid = grid / p.cgroups * p.cgroups;
    if (grid < p.cgroups) grid = p.cgroups;
    VinceProfScope prof(VINCE_TAG_XSTATS, (double)rows * (K + Co) * 2, stream);
    if (K == 64)
        hipLaunchKernelGGL((conv_xjoin_kernel<64, 3, false, 0, true>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((conv_xjoin_kernel<128, 2, false, 0, true>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

static int expand_dgrad_common(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out, int accumulate,
                               const uint8_t* acc_mask, const vince_bn_reduce* bnred, const uint8_t* out_mask, double* gsums,
                               int32_t replicas, void* stream);

extern "C" int vince_conv_expand_dgrad(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out,
                                       int accumulate, const uint8_t* acc_mask, const vince_bn_reduce* bnred, int32_t replicas,
                                       void* stream) {
    return expand_dgrad_common(dtype, dy, wt, rows, K, Co, out, accumulate, acc_mask, bnred, nullptr, nullptr, replicas, stream);
}

extern "C" int vince_conv_expand_dgrad_masked(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out,
                                              int accumulate, const uint8_t* acc_mask, const uint8_t* out_mask, double* gsums,
                                              int32_t replicas, void* stream) {
    VINCE_CHECK_ARG(((uintptr_t)out_mask & 7) == 0, VINCE_E_ALIGN, "vince_conv_expand_dgrad_masked: out_mask must be 8-byte aligned");
    return expand_dgrad_common(dtype, dy, wt, rows, K, Co, out, accumulate, acc_mask, nullptr, out_mask, gsums, replicas, stream);
}

static int expand_dgrad_common(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out, int accumulate,
                               const uint8_t* acc_mask, const vince_bn_reduce* bnred, const uint8_t* out_mask, double* gsums,
                               int32_t replicas, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv_expand_dgrad: bf16 only");
    VINCE_CHECK_ARG(dy && wt && out && rows > 0, VINCE_E_ARG, "vince_conv_expand_dgrad: null pointer");
    VINCE_CHECK_ARG(K == 64 || K == 128, VINCE_E_UNSUPPORTED, "vince_conv_expand_dgrad: K=%d (64 or 128)", K);
    VINCE_CHECK_ARG(Co > 0 && Co % XJ_CG == 0, VINCE_E_SHAPE, "vince_conv_expand_dgrad: Co=%d must be a multiple of %d", Co, XJ_CG);
    VINCE_CHECK_ARG(!acc_mask || accumulate, VINCE_E_ARG, "vince_conv_expand_dgrad: acc_mask needs accumulate");
    VINCE_CHECK_ARG(!bnred || !bnred->y || (bnred->mean && bnred->invstd && bnred->sums && !bnred->mask_scale), VINCE_E_ARG,
                    "vince_conv_expand_dgrad: bnred needs y, mean, invstd, sums (mask through mask_bits or none)");
    VINCE_CHECK_ARG((((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)out) & 15) == 0 && ((uintptr_t)acc_mask & 7) == 0, VINCE_E_ALIGN,
                    "vince_conv_expand_dgrad: pointers must be 16-byte (acc_mask 8-byte) aligned");
    const unsigned long long xb = (unsigned long long)rows * K * 2, wb = (unsigned long long)Co * K * 2;
    VINCE_CHECK_ARG(xb < 0x7ff00000ull && rows < (1ll << 31), VINCE_E_UNSUPPORTED, "vince_conv_expand_dgrad: input beyond the 31-bit buffer offsets");
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    XjParams p;
    memset(&p, 0, sizeof(p));
    p.x = dy; p.w = wt; p.out = out; p.identity = out; p.replicas = replicas;
    p.relu = accumulate ? 1 : 0;
    p.acc_mask = acc_mask;
    if (bnred && bnred->y) {
        p.br_y = bnred->y; p.br_bits = bnred->mask_bits; p.br_mean = bnred->mean; p.br_invstd = bnred->invstd; p.br_sums = bnred->sums;
    }
    p.out_mask = out_mask;
    if (gsums) p.br_sums = gsums;
    p.rows = (uint32_t)rows; p.Co = (uint32_t)Co; p.x_bytes = (uint32_t)xb; p.w_bytes = (uint32_t)wb;
    p.ptiles = (int)((rows + XJ_PX - 1) / XJ_PX);
    p.cgroups = Co / XJ_CG;
    long grid = xj_num_cu();
    const long items = (long)p.ptiles * p.cgroups;
    if (grid > items) grid = items;
    grid = grid / p.cgroups * p.cgroups;
    if (grid < p.cgroups) grid = p.cgroups;
    VinceProfScope prof(VINCE_TAG_XDGRAD, (double)rows * 2 * (K + Co * (1 + (accumulate ? 1 : 0) + (p.br_y ? 1 : 0))) +
                        (double)rows * Co / 8 * ((acc_mask ? 1 : 0) + (p.br_bits ? 1 : 0) + (out_mask ? 1 : 0)), stream);
    if (K == 64)
        hipLaunchKernelGGL((conv_xjoin_kernel<64, 3, false, 0, false, true>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((conv_xjoin_kernel<128, 2, false, 0, false, true>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

static int expand_join_common(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                              const float* out_scale, const float* out_shift, const void* identity,
                              const float* id_scale, const float* id_shift, void* out, void* y_raw, uint8_t* mask_out,
                              int relu, const void* w_next, int32_t co_next, void* y_next, double* stats_next, int32_t replicas_next,
                              const float* bias_next, int relu_next, void* stream);

extern "C" int vince_conv_expand_join(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                                      const float* out_scale, const float* out_shift, const void* identity,
                                      const float* id_scale, const float* id_shift, void* out, void* y_raw, uint8_t* mask_out,
                                      int relu, void* stream) {
    return expand_join_common(dtype, x, w, rows, K, Co, out_scale, out_shift, identity, id_scale, id_shift, out, y_raw, mask_out, relu,
                              nullptr, 0, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int vince_conv_expand_join_next(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                                           const float* out_scale, const float* out_shift, const void* identity,
                                           const float* id_scale, const float* id_shift, void* out, void* y_raw, uint8_t* mask_out,
                                           int relu, const void* w_next, int32_t Co_next, void* y_next, double* stats_next,
                                           int32_t replicas_next, const float* bias_next, int relu_next, void* stream) {
    VINCE_CHECK_ARG(w_next && y_next, VINCE_E_ARG, "vince_conv_expand_join_next: null pointer");
    VINCE_CHECK_ARG(K == 64 && Co == XJ_K2 && (Co_next == 64 || (Co_next == 128 && !id_scale)), VINCE_E_UNSUPPORTED,
                    "vince_conv_expand_join_next: K=%d Co=%d Co_next=%d (64, 256, 64 or -- plain identity only -- 128: layer1)", K, Co, Co_next);
    VINCE_CHECK_ARG((((uintptr_t)w_next | (uintptr_t)y_next) & 15) == 0, VINCE_E_ALIGN, "vince_conv_expand_join_next: pointers must be 16-byte aligned");
    return expand_join_common(dtype, x, w, rows, K, Co, out_scale, out_shift, identity, id_scale, id_shift, out, y_raw, mask_out, relu,
                              w_next, Co_next, y_next, stats_next, replicas_next, bias_next, relu_next, stream);
}

static int expand_join_common(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                              const float* out_scale, const float* out_shift, const void* identity,
                              const float* id_scale, const float* id_shift, void* out, void* y_raw, uint8_t* mask_out,
                              int relu, const void* w_next, int32_t co_next, void* y_next, double* stats_next, int32_t replicas_next,
                              const float* bias_next, int relu_next, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv_expand_join: bf16 only (fp32 runs vince_conv_igemm's join epilogue)");
    VINCE_CHECK_ARG(x && w && out_scale && out_shift && identity && out && rows > 0, VINCE_E_ARG, "vince_conv_expand_join: null pointer");
    VINCE_CHECK_ARG(K == 64 || K == 128, VINCE_E_UNSUPPORTED, "vince_conv_expand_join: K=%d (64 or 128)", K);
    VINCE_CHECK_ARG(Co > 0 && Co % XJ_CG == 0, VINCE_E_SHAPE, "vince_conv_expand_join: Co=%d must be a multiple of %d", Co, XJ_CG);
    VINCE_CHECK_ARG(!id_scale == !id_shift, VINCE_E_ARG, "vince_conv_expand_join: id_scale and id_shift come together");
    VINCE_CHECK_ARG(!y_raw || mask_out, VINCE_E_ARG, "vince_conv_expand_join: y_raw comes with mask_out (the training forward)");
    VINCE_CHECK_ARG(!mask_out || (out != identity && (((uintptr_t)y_raw & 15) | ((uintptr_t)mask_out & 7)) == 0), VINCE_E_ARG,
                    "vince_conv_expand_join: saving needs out != identity, y_raw 16-byte and mask_out 8-byte aligned");
    VINCE_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)identity | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN,
                    "vince_conv_expand_join: pointers must be 16-byte aligned");
    const unsigned long long xb = (unsigned long long)rows * K * 2, wb = (unsigned long long)Co * K * 2;
    VINCE_CHECK_ARG(xb < 0x7ff00000ull && rows < (1ll << 31), VINCE_E_UNSUPPORTED, "vince_conv_expand_join: input beyond the 31-bit buffer offsets");
    XjParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.identity = identity; p.out = out; p.y_raw = y_raw; p.mask_out = mask_out;
    p.out_scale = out_scale; p.out_shift = out_shift; p.id_scale = id_scale; p.id_shift = id_shift;
    p.rows = (uint32_t)rows; p.Co = (uint32_t)Co; p.x_bytes = (uint32_t)xb; p.w_bytes = (uint32_t)wb;
    p.ptiles = (int)((rows + XJ_PX - 1) / XJ_PX);
    p.cgroups = Co / XJ_CG;
    p.relu = relu;
    p.w2 = w_next; p.y2 = y_next; p.stats2 = stats_next; p.w2_bytes = (uint32_t)co_next * XJ_K2 * 2; p.bias2 = bias_next; p.relu2 = relu_next;
    p.replicas2 = (replicas_next <= 0 || replicas_next > VINCE_STATS_REPLICAS) ? VINCE_STATS_REPLICAS : replicas_next;
    const int n_cu = xj_num_cu();
    static const int wg_per_cu = VINCE_MEASURE_KNOB("xj_wgs", 1);   // (measurement aid)
    long grid = (long)n_cu * wg_per_cu;
    const long items = (long)p.ptiles * p.cgroups;
    if (grid > items) grid = items;
    grid = grid / p.cgroups * p.cgroups;                    // every workgroup keeps one channel group
    if (grid < p.cgroups) grid = p.cgroups;
#define VINCE_XJ_LAUNCH(KK, SS, AA, SV) \
    hipLaunchKernelGGL((conv_xjoin_kernel<KK, SS, AA, SV>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p)
#define VINCE_XJ_PICK(KK, SS)                                                                             \
    do {                                                                                                  \
        if (y_raw) { if (id_scale) VINCE_XJ_LAUNCH(KK, SS, true, 2); else VINCE_XJ_LAUNCH(KK, SS, false, 2); }            \
        else if (mask_out) { if (id_scale) VINCE_XJ_LAUNCH(KK, SS, true, 1); else VINCE_XJ_LAUNCH(KK, SS, false, 1); }    \
        else { if (id_scale) VINCE_XJ_LAUNCH(KK, SS, true, 0); else VINCE_XJ_LAUNCH(KK, SS, false, 0); }                  \
    } while (0)
#define VINCE_XJ_LAUNCH_NEXT(AA, SV) \
    hipLaunchKernelGGL((conv_xjoin_kernel<64, 3, AA, SV, false, false, 64>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p)
#define VINCE_XJ_LAUNCH_NEXT128(SV) \
    hipLaunchKernelGGL((conv_xjoin_kernel<64, 2, false, SV, false, false, 128>), dim3((unsigned)grid), dim3(XJ_THREADS), 0, (hipStream_t)stream, p)
    VinceProfScope prof(VINCE_TAG_XJOIN, (double)rows * 2 * (K + Co * (2 + (y_raw ? 1 : 0)) + co_next) + (mask_out ? (double)rows * Co / 8 : 0), stream);
    if (w_next && co_next == 128) {
        if (y_raw) VINCE_XJ_LAUNCH_NEXT128(2); else if (mask_out) VINCE_XJ_LAUNCH_NEXT128(1); else VINCE_XJ_LAUNCH_NEXT128(0);
    } else if (w_next) {
        if (y_raw) { if (id_scale) VINCE_XJ_LAUNCH_NEXT(true, 2); else VINCE_XJ_LAUNCH_NEXT(false, 2); }
        else if (mask_out) { if (id_scale) VINCE_XJ_LAUNCH_NEXT(true, 1); else VINCE_XJ_LAUNCH_NEXT(false, 1); }
        else { if (id_scale) VINCE_XJ_LAUNCH_NEXT(true, 0); else VINCE_XJ_LAUNCH_NEXT(false, 0); }
    } else if (K == 64) VINCE_XJ_PICK(64, 3); else VINCE_XJ_PICK(128, 2);
#undef VINCE_XJ_LAUNCH_NEXT
#undef VINCE_XJ_LAUNCH_NEXT128
#undef VINCE_XJ_PICK
#undef VINCE_XJ_LAUNCH
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
