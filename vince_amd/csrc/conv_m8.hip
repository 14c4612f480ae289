// The 8-wavefront implicit-GEMM core for the MFMA-bound convolutions (bf16): 256 pixels x 256 output channels per 512-thread
// workgroup, one workgroup per CU, K tile = 64 elements (whole 128-byte lines per gathered row).
//
// Replaces, for the long-reduction layers, the same reference call sites as conv_igemm.hip: the 3x3 and the wide 1x1 convolutions
// of layer3 / layer4 (models/building_blocks/resnet.py:34-50,107-135) and their input gradients.  Same operation, same epilogue
// (conv_core.h), same summation order per output element (K tiles ascending, 16-element MFMA steps ascending), so the results are
// BIT-IDENTICAL to conv_igemm's.
//
// Why a second core: conv_igemm's 4-wavefront tiles gather 64-byte row pieces (request-bound, tools/micro/feed_micro) and run
// their load / LDS-read / MFMA phases in lock-step.  Here
//   * operand bytes per FLOP are half those of the 128 x 128 tile, every gathered piece is a whole cache line;
//   * the 8 wavefronts form two groups of four (one wavefront of each group per SIMD) that run ONE BARRIER APART: while one group
//     issues its 8 MFMAs of a quadrant (256 cycles of the matrix pipe) the other reads the fragments of its next quadrant from LDS
//     and issues its share of the LDS-DMA -- the matrix pipe of a SIMD always has one wavefront feeding it;
//   * the LDS-DMA runs 6 phases (1.5 K tiles, 96 KB per CU) ahead of its consumer with counted vmcnt, never drained in the loop.
//
// Tile bookkeeping.  Wavefront w = (wm, wn) = (w >> 2, w & 3) owns pixels [wm*128, +128) x channels [wn*64, +64) of the tile as
// acc[c][i]: channel half c (32 channels), pixel tile i (32 pixels); a QUADRANT is one channel half x one pixel half (2 pixel
// tiles) x the 4 MFMA steps of a K tile = 8 x v_mfma_f32_32x32x16_bf16.  A K tile lives in LDS as four HALF-TILES of 128 rows x
// 128 bytes: X0 / X1 (pixel half ph of every wm: row r = wm*64 + (pixel & 63)) and C0 / C1 (channel half c of every wn: row
// r = wn*32 + (channel & 31)); two K-tile slots = 8 half-tile buffers = 128 KB.  Rows are unpadded (LDS-DMA writes lane-linear);
// the 16-byte slot of logical K chunk q of row r is q ^ ((r >> 1) & 7) (source-side swizzle, conflict-free ds_read_b128).
//
// Schedule (phase = quadrant; 4 phases per K tile, the loop body holds 2 K tiles so that the register roles are static):
//   phase 4t+0: read X0(t) -> Xr       issue X1(t+1)    MFMA (c0, p0)     c0(t) already sits in a W register set
//   phase 4t+1: read C1(t) -> Wother   issue C0(t+2)    MFMA (c1, p0)
//   phase 4t+2: read X1(t) -> Xr       issue X0(t+2)    MFMA (c1, p1)
//   phase 4t+3: read C0(t+1) -> Wother issue C1(t+2)    MFMA (c0, p1)
// Each phase is   reads ; DMA issue ; s_waitcnt vmcnt(10) ; s_barrier ; MFMA cluster ; s_barrier.
// Correctness of the overlap (every wavefront issues 2 DMA instructions per phase, for the half-tile read 6 phases later):
//   RAW  the half-tile issued in phase P is first read in phase P+6 of group 0, which starts behind barrier #(2P+13) (group 0's
//        phase P spans barriers #(2P+1)..#(2P+3), group 1's #(2P+2)..#(2P+4)).  vmcnt(10) at the end of the read slot of phase P+5
//        leaves only the DMAs of phases P+1..P+5 in flight, and that wait precedes barrier #(2P+12) in group 0 and #(2P+13) in
//        group 1: every wavefront's share has landed before any wavefront reads it.
//   WAR  the half-tile overwritten in phase P was last read in phase P-2: group 1's reads of phase P-2 are retired by its
//        lgkmcnt(0) before its MFMA cluster, i.e. before barrier #(2P), and the earliest DMA issue of phase P is behind #(2P+1).
#include <stdlib.h>
#include <string.h>

#include "conv_core.h"

namespace {

constexpr int M8_PT = 256, M8_CT = 256;
constexpr int M8_HT = 128 * 128;        // half-tile bytes
constexpr int M8_SLOT = 4 * M8_HT;      // X0 X1 C0 C1
constexpr int M8_CRS = M8_CT * 2 + 16;  // epilogue tile row stride
constexpr int M8_EPI = M8_PT * M8_CRS + 8 * 32 * 8 * 2 * 4;
constexpr int M8_LDS = 2 * M8_SLOT > M8_EPI ? 2 * M8_SLOT : M8_EPI;

#ifndef M8_ILV
#define M8_ILV 0      // 1: DMA issue between the MFMAs of the quadrant (measured slower: 68.4 vs 62.5 us on layer3's 3x3); 0: in the read slot
#endif

#define M8_FENCE()                               \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

// Measurement build only (-DVINCE_MEASURE, VINCE_M8_ABLATE): 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no barriers,
// 16 every DMA zero-filled (same instructions, no memory traffic), 32 no epilogue, 64 no s_setprio around the MFMA clusters.  Results are garbage under any of them.
#ifdef VINCE_MEASURE
#define M8_ABL(bit) (p.ablate & (bit))
#else
#define M8_ABL(bit) 0
#endif

// One quadrant: 8 MFMAs.  `between(m)` runs after MFMA m: the wavefront's share of the LDS-DMA (address step + two issues) rides in
// the issue slots the matrix pipe leaves free (an MFMA occupies the pipe for 32 cycles and the issue port for a few), so the read
// slot of the partner group carries nothing but its ds_reads.
template <int C, int PH, typename F>
__device__ __forceinline__ void m8_quad(const ConvParams& p, const uint4 (&wr)[4], const uint4 (&xr)[2][4], f32x16_t (&acc)[2][4],
                                        F&& between) {
    if (M8_ABL(2)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(wr[s].x), "v"(wr[s].w), "v"(xr[0][s].x), "v"(xr[0][s].w), "v"(xr[1][s].x), "v"(xr[1][s].w));
        between(0); between(2); between(4);
        return;
    }
    if (!M8_ABL(64)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Mma<bf16_t>::run(wr[s], xr[i][s], acc[C][2 * PH + i]);
            if (M8_ILV && (s * 2 + i) % 2 == 0 && s * 2 + i < 6) {
                __builtin_amdgcn_sched_barrier(0);
                between(s * 2 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    if (!M8_ABL(64)) __builtin_amdgcn_s_setprio(0);
}

template <int MODE>
__global__ __launch_bounds__(512) void conv_m8_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[M8_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ptile = tile / p.ctiles, ctile = tile - ptile * p.ctiles;
    const int p0 = ptile * M8_PT, c0 = ctile * M8_CT;
    const vince_conv_desc& d = p.d;
    constexpr uint32_t OOB = 0x80000000u;

    const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
    const v4i_t rsrc_w = make_rsrc(p.w, p.w_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    // ---- DMA assignment: instruction e (0, 1) of this wavefront fills half-tile rows (e*8 + wave)*8 .. +7; lane -> row + (lane >> 3),
    // 16-byte slot (lane & 7), logical K chunk slot ^ ((row >> 1) & 7)
    const int drow = lane >> 3;
    int hb[2][2], wb[2][2];       // [pixel half][e]
    uint32_t nb[2][2];
    uint32_t wrow[2][2];          // [channel half][e]: element offset of the weight row, OOB past Co
    uint32_t clog2[2];            // per e: byte offset of this lane's logical chunk inside the 128-byte K row
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = (e * 8 + wave) * 8 + drow;                 // half-tile row 0..127
        clog2[e] = (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const uint32_t m = p0 + (r >> 6) * 128 + ph * 64 + (r & 63);
            const bool rv = m < (uint32_t)p.M;
            const uint32_t mm = rv ? m : 0;
            const uint32_t n = fastdiv(mm, p.div_howo);
            const uint32_t rem = mm - n * p.div_howo.d;
            const uint32_t ho = fastdiv(rem, p.div_wo);
            const uint32_t wo = rem - ho * p.div_wo.d;
            hb[ph][e] = rv ? (int)(ho * d.sh) : -0x40000000;     // an invalid row fails every bounds test
            wb[ph][e] = wo * d.sw;
            nb[ph][e] = n * (uint32_t)(d.Hi * d.Wi);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int co = c0 + (r >> 5) * 64 + c * 32 + (r & 31);
            wrow[c][e] = co < d.Co ? (uint32_t)co * (uint32_t)(d.WT * d.Ci) : OOB;
        }
    }
    // Byte offsets of this lane's two rows per half-tile kind, valid for the K tile the kind issues next.  Every kind walks the K
    // tiles in order, so a step is "offset += 128"; the tap decode (bounds tests, multiplies) runs once per tap and kind, in a
    // wave-uniform branch.  Rows past the image / the tile tail / the end of the reduction sit at OOB and stay there (the buffer
    // descriptor zero-fills them).
    uint32_t xoff[2][2], woff[2][2];
    auto retap = [&](const int kind, const int kt, const bool in_loop) {
        const bool live = kt < p.nkt && !(in_loop && M8_ABL(16));
        const int tap = live ? (kt >> p.log2_ktpt) : 0;
        const int a = (int)(((uint32_t)tap * p.tb_mul) >> 16), b = tap - a * d.TB;
        if (kind < 2) {
            const int dh = d.dh0 + a * d.dhs, dw = d.dw0 + b * d.dws;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int hi = hb[kind][e] + dh, wi = wb[kind][e] + dw;
                const bool ok = live && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
                xoff[kind][e] = ok ? (nb[kind][e] + (uint32_t)(hi * d.Wi + wi)) * (uint32_t)(d.Ci * 2) + clog2[e] : OOB;
            }
        } else {
            const int widx = d.wt0 + a * d.wta + b * d.wtb;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t row = wrow[kind - 2][e];
                woff[kind - 2][e] = (live && row < OOB) ? (row + (uint32_t)(widx * d.Ci)) * 2u + clog2[e] : OOB;
            }
        }
    };
    // the DMA of half-tile `kind` (0 X0, 1 X1, 2 C0, 3 C1) of K tile kt into slot (kt & 1), in three steps: 0 = address step (tap
    // decode at a tap boundary), 2 / 4 = the two instructions
    auto issue_step = [&](const int step, const int kind, const int kt, const int slot, const bool in_loop = true) {
        if (in_loop && M8_ABL(1)) return;
        if (step == 0) {
            if ((kt & p.ktpt_mask) == 0) retap(kind, kt, in_loop);
            return;
        }
        const int e = step == 2 ? 0 : 1;
        const uint32_t lds = __builtin_amdgcn_readfirstlane(smem_base + slot * M8_SLOT + kind * M8_HT + wave * 1024 + e * 8192);
        if (kind < 2) {
            lds_dma16_m0(lds, xoff[kind][e], rsrc_x);
            xoff[kind][e] += 128;
        } else {
            lds_dma16_m0(lds, woff[kind - 2][e], rsrc_w);
            woff[kind - 2][e] += 128;
        }
    };
    auto issue = [&](const int kind, const int kt, const int slot, const bool in_loop = true) {
        issue_step(0, kind, kt, slot, in_loop);
        issue_step(2, kind, kt, slot, in_loop);
        issue_step(4, kind, kt, slot, in_loop);
    };

    // ---- fragment read offsets: row (lane & 31) of a 32-row MFMA tile, K step s -> chunk (2s + (lane >> 5)) ^ swizzle(row)
    int foff[4];
    {
        const int sw = ((lane & 31) >> 1) & 7, kh = lane >> 5;
#pragma unroll
        for (int s = 0; s < 4; ++s) foff[s] = (lane & 31) * 128 + (((2 * s + kh) ^ sw) << 4);
    }
    const unsigned char* xbase = smem + wm * 64 * 128;
    const unsigned char* wbase = smem + 2 * M8_HT + wn * 32 * 128;
    auto read_x = [&](const int slot, const int ph, uint4 (&xr)[2][4]) {
        if (M8_ABL(4)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) xr[i][s] = *(const uint4*)(xbase + slot * M8_SLOT + ph * M8_HT + i * 4096 + foff[s]);
    };
    auto read_w = [&](const int slot, const int c, uint4 (&wr)[4]) {
        if (M8_ABL(4)) return;
#pragma unroll
        for (int s = 0; s < 4; ++s) wr[s] = *(const uint4*)(wbase + slot * M8_SLOT + c * M8_HT + foff[s]);
    };

    f32x16_t acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][i][e] = 0.f;

    // ---- prologue: the seven half-tiles the first six phases (and the pre-read of C0(0)) consume
    issue(2, 0, 0, false);
    issue(0, 0, 0, false);
    issue(3, 0, 0, false);
    issue(1, 0, 0, false);
    issue(2, 1, 1, false);
    issue(0, 1, 1, false);
    issue(3, 1, 1, false);
    wait_vmcnt<10>();                       // C0(0), X0(0) have landed (this wavefront's share)
    M8_FENCE();
    __builtin_amdgcn_s_barrier();
    M8_FENCE();
    uint4 wa[4] = {}, wbq[4] = {}, xr[2][4] = {};
    read_w(0, 0, wa);
    M8_FENCE();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0
    M8_FENCE();

    // phase: reads ; [DMA issue] ; vmcnt ; barrier ; MFMA quadrant [with the DMA issue between the MFMAs] ; barrier.
    // vmcnt: the half-tile read in the NEXT phase was issued 5 phases before this one's issue -- 5 x 2 instructions may stay in flight
    // when this phase's issue precedes the wait (M8_ILV 0), 4 x 2 when it follows it (M8_ILV 1).
#define M8_PHASE(READ, KIND, KT, SLOT, C, PH, WR)                                                         \
    do {                                                                                                  \
        READ;                                                                                             \
        if (!M8_ILV) issue(KIND, KT, SLOT);                                                               \
        wait_vmcnt<(M8_ILV ? 8 : 10)>();                                                                  \
        M8_FENCE();                                                                                       \
        if (!M8_ABL(8)) __builtin_amdgcn_s_barrier();                                                     \
        M8_FENCE();                                                                                       \
        m8_quad<C, PH>(p, WR, xr, acc, [&](const int m) { if (M8_ILV) issue_step(m, KIND, KT, SLOT); });  \
        M8_FENCE();                                                                                       \
        if (!M8_ABL(8)) __builtin_amdgcn_s_barrier();                                                     \
        M8_FENCE();                                                                                       \
    } while (0)

    for (int kt = 0; kt < p.nkt; kt += 2) {
        // even K tile (slot 0): c0 in wa
        M8_PHASE(read_x(0, 0, xr), 1, kt + 1, 1, 0, 0, wa);
        M8_PHASE(read_w(0, 1, wbq), 2, kt + 2, 0, 1, 0, wbq);
        M8_PHASE(read_x(0, 1, xr), 0, kt + 2, 0, 1, 1, wbq);
        M8_PHASE(read_w(1, 0, wbq), 3, kt + 2, 0, 0, 1, wa);
        // odd K tile (slot 1): c0 in wbq
        M8_PHASE(read_x(1, 0, xr), 1, kt + 2, 0, 0, 0, wbq);
        M8_PHASE(read_w(1, 1, wa), 2, kt + 3, 1, 1, 0, wa);
        M8_PHASE(read_x(1, 1, xr), 0, kt + 3, 1, 1, 1, wa);
        M8_PHASE(read_w(0, 0, wa), 3, kt + 3, 1, 0, 1, wbq);
    }
#undef M8_PHASE
    if (wm == 0) __builtin_amdgcn_s_barrier();   // re-align the two groups
    wait_vmcnt<0>();                             // the zero fills issued past the end
    __syncthreads();

    if (M8_ABL(32)) {
        float t = 0.f;
        for (int c = 0; c < 2; ++c) for (int i = 0; i < 4; ++i) t += acc[c][i][0];
        if (t == 1.2345f) ((float*)p.out)[0] = t;
        return;
    }
    conv_epilogue<bf16_t, M8_CT, M8_CRS, MODE, M8_PT, 4, 512, 4>(p, smem, acc, tile, p0, c0, tid, lane, wave, wm, wn);
}

}  // namespace

// Launch on the 8-wavefront core.  Returns VINCE_OK, or -1 when the shape does not qualify (the caller falls back to conv_igemm's
// own tiles).  mode: 0 forward, 1 gradient epilogues, 2 forward residual join.
int vince_conv_m8_launch(vince_conv::ConvParams& p, int mode, hipStream_t stream) {
    const vince_conv_desc& d = p.d;
    if (!(p.in_bytes && p.w_bytes) || d.Cs != 0 || d.Ci % 64 != 0 || p.kt_per_split != 0) return VINCE_M8_NOT_ELIGIBLE;
    const int taps = d.TA * d.TB;
    p.kt_per_tap = d.Ci / 64;
    if (p.kt_per_tap & (p.kt_per_tap - 1)) return VINCE_M8_NOT_ELIGIBLE;      // the per-tap walk wants a power of two (every ResNet layer)
    p.ktpt_mask = p.kt_per_tap - 1;
    p.log2_ktpt = 0;
    while ((1 << p.log2_ktpt) < p.kt_per_tap) ++p.log2_ktpt;
    p.nkt = taps * p.kt_per_tap;
    p.ptiles = (p.M + M8_PT - 1) / M8_PT;
    p.ctiles = (d.Co + M8_CT - 1) / M8_CT;
    p.variant = 3;
#ifdef VINCE_MEASURE
    p.ablate = VINCE_MEASURE_KNOB("m8_ablate", 0);
#endif
    const dim3 grid(p.ptiles * p.ctiles), block(512);
    if (mode == 0) hipLaunchKernelGGL(conv_m8_kernel<0>, grid, block, 0, stream, p);
    else if (mode == 1) hipLaunchKernelGGL(conv_m8_kernel<1>, grid, block, 0, stream, p);
    else hipLaunchKernelGGL(conv_m8_kernel<2>, grid, block, 0, stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
