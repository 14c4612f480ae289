// Generalised tap convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces the reference's conv2d / nn.Linear call sites (models/building_blocks/resnet.py:34-50,170;
// models/vince_model.py:38-42) and their input-gradient halves.  See include/vince_hip.h for the op definition.
//
// Mapping to the hardware
//   GEMM:  D[co][pix] = sum_k  W[co][k] * X[pix][k],   k = (tap, ci) with ci fastest -> both operands are
//          K-contiguous in HBM (weights [Co][T][Ci], activations NHWC), so every global access is a 16-byte
//          chunk (8 bf16 / 4 f32) and a wavefront reads whole 64-byte row segments.
//   Tile:  128 pixels x CT (64|128) output channels per 256-thread workgroup (4 waves as 2x2), K step = 64 bytes.
//          Each wave owns 64 pixels x CT/2 channels as 32x32 MFMA tiles: v_mfma_f32_32x32x16_bf16 (bf16) or
//          4 x v_mfma_f32_32x32x2_f32 per 16-byte fragment (exact fp32).  A and B fragments are the same 16 bytes
//          per lane at row (lane&31), k-offset (lane>>5)*16 B, so the dtype only changes the MFMA, not the loads.
//   LDS:   register-staged double buffer (global -> VGPR -> ds_write_b128), rows padded 64 -> 80 B so the
//          ds_read_b128 fragment reads of 16 consecutive rows land on 16 distinct 16-B slots (conflict free).
//   Store: accumulators hold one pixel per lane and 4 consecutive channels per register group; the tile is
//          transposed through LDS and written as full 16-byte chunks along the channel axis (coalesced NHWC rows).
//          The optional BatchNorm statistics (per-channel sum / sum of squares of the values just stored) are
//          reduced in registers -> shuffles -> LDS -> one fp64 atomic per channel per workgroup.
//   Grid:  1-D, remapped so that each XCD (private L2) owns a contiguous range of tiles; channel tiles of the
//          same pixel tile are adjacent and re-read the activation tile from that L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#include "conv_core.h"

namespace {

constexpr int PT = 128;   // pixels per workgroup tile

// Measurement build only (-DVINCE_MEASURE, VINCE_CONV_ABLATE): 1 no DMA, 2 no MFMA, 4 no barrier, 8 no statistics atomics, 16 no main
// loop, 32 no output stores, 64 launch only, 128 no epilogue.  The product library compiles none of it.
#ifdef VINCE_MEASURE
#define IG_ABL(bit) (p.ablate & (bit))
#else
#define IG_ABL(bit) 0
#endif

// Wavefront priority around an MFMA cluster (VINCE_MFMA_PRIO, build-time): with equal priorities the SIMD's arbiter interleaves the
// MFMAs of its resident wavefronts, which locks them into the same phase (all in their MFMA cluster, then all in their
// LDS / barrier phase); a raised priority lets one wavefront run its cluster through while the other fetches.
#ifndef VINCE_MFMA_PRIO
#define VINCE_MFMA_PRIO 0
#endif
template <int ON> __device__ __forceinline__ void mfma_prio() {
#if VINCE_MFMA_PRIO
    if constexpr (ON) __builtin_amdgcn_s_setprio(VINCE_MFMA_PRIO); else __builtin_amdgcn_s_setprio(0);
#endif
}

template <typename T, int CT, int KC>
struct Smem {
    static constexpr int RS = KC * 16 + 16;    // LDS row stride: KC 16-byte K chunks + one pad chunk (odd multiple of 16 B)
    static constexpr int MAIN = 2 * (CT + PT) * RS;
    static constexpr int CRS = CT * (int)sizeof(T) + 16;   // epilogue tile row stride (bytes)
    static constexpr int EPI = PT * CRS + 4 * CT * 2 * 4;   // + statistics scratch
    static constexpr int BYTES = MAIN > EPI ? MAIN : EPI;
};


template <typename T, int CT, int KC, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int CJ = CT / 64;          // 32-channel MFMA tiles per wave
    constexpr int RSTEP = 256 / KC;      // rows covered by one staging pass
    constexpr int XROWS = PT / RSTEP, WROWS = CT / RSTEP;
    constexpr int RS = Smem<T, CT, KC>::RS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Smem<T, CT, KC>::BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ptile = tile / p.ctiles, ctile = tile - ptile * p.ctiles;
    const int p0 = ptile * PT, c0 = ctile * CT;
    const vince_conv_desc& d = p.d;

    // ---- per-thread staging assignment: chunk column cj of rows r + e*RSTEP ------------------------------
    const int cj = tid % KC, r = tid / KC;
    int hb[XROWS], wb[XROWS];
    size_t nb[XROWS];
    bool rv[XROWS];
#pragma unroll
    for (int e = 0; e < XROWS; ++e) {
        uint32_t m = p0 + r + e * RSTEP;
        rv[e] = m < (uint32_t)p.M;
        uint32_t mm = rv[e] ? m : 0;
        uint32_t n = fastdiv(mm, p.div_howo);
        uint32_t rem = mm - n * p.div_howo.d;
        uint32_t ho = fastdiv(rem, p.div_wo);
        uint32_t wo = rem - ho * p.div_wo.d;
        hb[e] = ho * d.sh;
        wb[e] = wo * d.sw;
        nb[e] = (size_t)n * d.Hi * d.Wi;
    }
    const T* __restrict__ in = (const T*)p.in;
    const T* __restrict__ wgt = (const T*)p.w;

    uint4 xr[XROWS], wr[WROWS];
    auto load_tile = [&](int kt) {
        const int q = kt * KC + cj;
        const int tap = q >> p.log2_cpt, cc = q & p.cpt_mask;
        const int a = (int)(((uint32_t)tap * p.tb_mul) >> 16), b = tap - a * d.TB;
        const int dh = d.dh0 + a * d.dhs, dw = d.dw0 + b * d.dws;
        const int widx = d.wt0 + a * d.wta + b * d.wtb;
        const bool qv = q < p.total_chunks;
#pragma unroll
        for (int e = 0; e < XROWS; ++e) {
            const int hi = hb[e] + dh, wi = wb[e] + dw;
            const bool ok = rv[e] && qv && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *(const uint4*)(in + (nb[e] + (size_t)hi * d.Wi + wi) * p.cs + (size_t)cc * CH);
            xr[e] = v;
        }
#pragma unroll
        for (int e = 0; e < WROWS; ++e) {
            const int co = c0 + r + e * RSTEP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qv && co < d.Co) v = *(const uint4*)(wgt + ((size_t)co * d.WT + widx) * d.Ci + (size_t)cc * CH);
            wr[e] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* ws = smem + buf * (CT + PT) * RS;
        unsigned char* xs = ws + CT * RS;
#pragma unroll
        for (int e = 0; e < XROWS; ++e) *(uint4*)(xs + (r + e * RSTEP) * RS + cj * 16) = xr[e];
#pragma unroll
        for (int e = 0; e < WROWS; ++e) *(uint4*)(ws + (r + e * RSTEP) * RS + cj * 16) = wr[e];
    };

    f32x16_t acc[CJ][2];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frag_off = (lane & 31) * RS + (lane >> 5) * 16;
    for (int kt = 0; kt < p.nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < p.nkt) load_tile(kt + 1);
        const unsigned char* ws = smem + buf * (CT + PT) * RS + (wc * (CT / 2)) * RS + frag_off;
        const unsigned char* xs = smem + buf * (CT + PT) * RS + CT * RS + (wp * 64) * RS + frag_off;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            uint4 wf[CJ], xf[2];
#pragma unroll
            for (int j = 0; j < CJ; ++j) wf[j] = *(const uint4*)(ws + j * 32 * RS + s * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = *(const uint4*)(xs + i * 32 * RS + s * 32);
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
        }
        if (kt + 1 < p.nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<T, CT, Smem<T, CT, KC>::CRS, MODE>(p, smem, acc, tile, p0, c0, tid, lane, wave, wp, wc);
}


// ---------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (long reductions).  The K tile is 128 bytes per row and is filled by
// `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging VGPRs, no ds_write pass, loads of tile k+1 are in flight while
// tile k feeds the matrix cores, one barrier per tile.  An LDS-DMA instruction writes wave-uniform-base + lane*16,
// i.e. 8 rows x 128 B per wave instruction, so rows cannot be padded; bank conflicts of the ds_read_b128 fragment reads
// are removed instead by an XOR swizzle applied on the SOURCE side: the lane that fills 16-byte slot `pos` of row R
// fetches logical K chunk pos ^ ((R>>1)&7), and the fragment read of chunk c goes to slot c ^ ((R>>1)&7) -- 16
// consecutive rows then cover all 16 slots of the 256-byte bank window.  Out-of-image taps and tile tails are zero
// filled by the buffer descriptor's range check (offset forced past num_records).
template <typename T, int CT, int KC, int STAGES, int PTL = PT>
struct SmemD {
    static constexpr int KB = KC * 16;                       // bytes of K per row per stage
    static constexpr int XB = PTL * KB, WB = CT * KB, STAGE = XB + WB;
    static constexpr int MAIN = STAGES * STAGE;
    static constexpr int CRS = CT * (int)sizeof(T) + 16;
    static constexpr int EPI = PTL * CRS + 4 * CT * 2 * 4;
    static constexpr int BYTES0 = MAIN > EPI ? MAIN : EPI;
};

// K tile = KC 16-byte chunks per row; STAGES-deep LDS ring, prefetch distance STAGES-1 tiles, counted vmcnt so that the
// younger tiles stay in flight across the barrier (one barrier per K tile).
// PTL = pixels per workgroup tile (128 or 256).  The L2 -> LDS fill rate of a CU (measured ~19 B/clk with every CU
// streaming) caps a 128x128 tile at ~700 TFLOP/s chip-wide: 256 B of operands per K element feed 32768 FLOP.  The
// 256-pixel tile moves 25 % fewer bytes per FLOP (each wave owns 128 pixels x CT/2 channels).
template <typename T, int CT, int KC, int STAGES, int MINW, int PTL, int MODE, bool ROT = false>
__global__ __launch_bounds__(256, MINW) void conv_igemm_dlds_kernel(const ConvParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int CJ = CT / 64, PI = PTL / 64;
    using S = SmemD<T, CT, KC, STAGES, PTL>;
    constexpr int KB = S::KB;
    constexpr int RPW = 1024 / KB;                 // rows per wave DMA instruction (8 or 16)
    constexpr int RPP = 4 * RPW;                   // rows per pass of the 4 waves
    constexpr int XROWS = PTL / RPP, WROWS = CT / RPP;
    constexpr int PER_STAGE = XROWS + WROWS;       // DMA instructions per thread per stage
    constexpr int SWSH = KC == 8 ? 1 : 2, SWMASK = KC - 1;   // slot swizzle = (row >> SWSH) & SWMASK
    constexpr bool ILV = CT == 128 && STAGES == 3;    // DMA issue interleaved with the MFMAs (see issue_piece)
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::BYTES0];

    if (IG_ABL(64)) return;   // launch + workgroup dispatch only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wp = wave >> 1;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ptile = tile / p.ctiles, ctile = tile - ptile * p.ctiles;
    const int p0 = ptile * PTL, c0 = ctile * CT;
    const vince_conv_desc& d = p.d;
    constexpr uint32_t OOB = 0x80000000u;   // descriptors cover < 2 GiB, so this (and small increments of it) reads as zero

    const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
    // second input tensor of the LAST tap (vince_conv_epi.in2: a reduction split over two tensors); same descriptor otherwise
    const v4i_t rsrc_x2 = make_rsrc(p.in2 ? p.in2 : p.in, p.in2 ? p.in2_bytes : p.in_bytes);
    const int tap2 = p.in2 ? d.TA * d.TB - 1 : 0x7ffffff;
    bool src2 = false;   // (wave-uniform) the offsets in offx belong to the second tensor
    const v4i_t rsrc_w = make_rsrc(p.w, p.w_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    // lane -> (row r + RPP*e, slot cpos); the logical K chunk it fetches is cpos ^ swizzle(r), the same for every e
    const int cpos = tid % KC, r = tid / KC;
    const int c_log = cpos ^ ((r >> SWSH) & SWMASK);
    int hb[XROWS], wb[XROWS];
    uint32_t nb[XROWS];
    bool rv[XROWS];
#pragma unroll
    for (int e = 0; e < XROWS; ++e) {
        uint32_t m = p0 + r + e * RPP;
        rv[e] = m < (uint32_t)p.M;
        uint32_t mm = rv[e] ? m : 0;
        uint32_t n = fastdiv(mm, p.div_howo);
        uint32_t rem = mm - n * p.div_howo.d;
        uint32_t ho = fastdiv(rem, p.div_wo);
        uint32_t wo = rem - ho * p.div_wo.d;
        hb[e] = ho * d.sh;
        wb[e] = wo * d.sw;
        nb[e] = n * (uint32_t)(d.Hi * d.Wi);
    }

    // Per-lane byte offsets of the current K tile.  When a tap spans a whole number of K tiles (Ci*sizeof(T) multiple of
    // the tile width: every layer but the stem) all lanes change tap together, so between tap changes a K step is just
    // "offset += KB" -- the address arithmetic (tap decode, bounds tests, multiplies) runs once per tap, not per tile.
    uint32_t offx[XROWS], offw[WROWS];
    int cur_tap = -1;
    auto compute_offsets = [&](int q) {
        const int tap = q >> p.log2_cpt, cc = q & p.cpt_mask;
        const int a = (int)(((uint32_t)tap * p.tb_mul) >> 16), b = tap - a * d.TB;
        const int dh = d.dh0 + a * d.dhs, dw = d.dw0 + b * d.dws;
        const int widx = d.wt0 + a * d.wta + b * d.wtb;
        const bool qv = q < p.total_chunks;       // also false for kt >= nkt: the whole tile is zero filled
        src2 = __builtin_amdgcn_readfirstlane(tap) == tap2;    // (every lane of a tile is in the same tap here: uniform_taps)
        const uint32_t cs = src2 ? (uint32_t)p.cs2 : (uint32_t)p.cs;
#pragma unroll
        for (int e = 0; e < XROWS; ++e) {
            const int hi = hb[e] + dh, wi = wb[e] + dw;
            const bool ok = rv[e] && qv && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
            offx[e] = ok ? ((nb[e] + (uint32_t)(hi * d.Wi + wi)) * cs + (uint32_t)cc * CH) * (uint32_t)sizeof(T) : OOB;
        }
#pragma unroll
        for (int e = 0; e < WROWS; ++e) {
            const int co = c0 + r + e * RPP;
            const bool ok = qv && co < d.Co;
            offw[e] = ok ? (((uint32_t)co * (uint32_t)d.WT + (uint32_t)widx) * (uint32_t)d.Ci + (uint32_t)cc * CH) * (uint32_t)sizeof(T) : OOB;
        }
    };
    auto issue_tile = [&](int kt, int buf) {
        if (p.uniform_taps) {
            const int tap = kt >= p.nkt ? 0x7fffff : ((kt * KC) >> p.log2_cpt);   // wave-uniform
            if (tap != cur_tap) {
                compute_offsets(kt * KC + c_log);
                cur_tap = tap;
            } else {
#pragma unroll
                for (int e = 0; e < XROWS; ++e) offx[e] += KB;    // OOB (>= 2 GiB) stays out of range
#pragma unroll
                for (int e = 0; e < WROWS; ++e) offw[e] += KB;
            }
        } else {
            compute_offsets(kt * KC + c_log);
        }
        if constexpr (!ILV) {
            const uint32_t xs = __builtin_amdgcn_readfirstlane(smem_base + buf * S::STAGE + wave * 1024);
            const uint32_t ws = xs + S::XB;
#pragma unroll
            for (int e = 0; e < XROWS; ++e) lds_dma16(xs + e * 4096, offx[e], src2 ? rsrc_x2 : rsrc_x);
#pragma unroll
            for (int e = 0; e < WROWS; ++e) lds_dma16(ws + e * 4096, offw[e], rsrc_w);
        }
    };
    // ILV (the 3-stage, 128-channel configurations: long reductions): the PER_STAGE DMA instructions of the tile being prefetched
    // are issued one at a time BETWEEN the MFMAs of the current tile instead of in front of them -- a piece costs ~100-185 issue
    // cycles (cdna guide) that then hide under the matrix pipe.  Measured: 3x3 layers of layer2/3/4 -3..8 %; the 2-stage
    // short-reduction configurations lose 2-5 % and keep the up-front issue.
    auto issue_piece = [&](int piece, int buf) {
        const uint32_t xs = __builtin_amdgcn_readfirstlane(smem_base + buf * S::STAGE + wave * 1024);
        if (piece < XROWS) lds_dma16(xs + piece * 4096, offx[piece < XROWS ? piece : 0], src2 ? rsrc_x2 : rsrc_x);
        else lds_dma16(xs + S::XB + (piece - XROWS) * 4096, offw[piece >= XROWS ? piece - XROWS : 0], rsrc_w);
    };

    f32x16_t acc[CJ][PI];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < PI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // split-K (tiny-M GEMMs: the projection MLP): this workgroup reduces K tiles [kt0, kt1) only
    const int kt0 = p.kt_per_split > 0 ? (int)blockIdx.y * p.kt_per_split : 0;
    const int kt1 = IG_ABL(16) ? kt0 : (p.kt_per_split > 0 ? min(p.nkt, kt0 + p.kt_per_split) : p.nkt);   // ablate 16: no main loop
    // prologue: STAGES-1 tiles in flight (tiles past the end are issued as all-zero fills so the counts stay uniform)
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st) {
        issue_tile(kt0 + st, st);
        if constexpr (ILV) {
#pragma unroll
            for (int pc = 0; pc < PER_STAGE; ++pc) issue_piece(pc, st);
        }
    }
    wait_vmcnt<(STAGES - 2) * PER_STAGE>();
    __syncthreads();
    const int sw = ((lane & 31) >> SWSH) & SWMASK, khalf = lane >> 5;
    const int row_off = (lane & 31) * KB;
    int buf = 0, nbuf = STAGES - 1;
    if constexpr (ROT) {
        // ROTATED main loop.  A K tile is NP = KC/2 MFMA phases (16 reduction elements each).  The fragments of phase s+1 are
        // read from LDS BEFORE the MFMAs of phase s are issued, and the rotation carries across the tile boundary: the last
        // phase's MFMAs are issued AFTER the barrier that publishes the next tile, right behind the reads of that tile's first
        // fragments -- so every LDS round trip (and the barrier skew) hides under 8..16 MFMAs of the same wavefront instead
        // of relying on another resident workgroup.  Measured critical path of the plain loop on the 256x128 tile: ~850
        // cycles of ds_read latency + barrier per K tile next to 512 cycles of MFMA, not overlapped within a wavefront.
        //   iteration kt:  [reads F1(kt)] [MFMA F0(kt) + DMA pieces of tile kt+S-1] [vmcnt, barrier] [reads F0(kt+1)] [MFMA F1(kt) + rest]
        // DMA pieces issued before the wait (PB of them) stay in flight across it; the buffer they overwrite (tile kt-1's) was
        // released by the previous iteration's barrier, which every wavefront reaches with its fragment reads complete.
        constexpr int NP = KC / 2;
        static_assert(NP % 2 == 0, "two fragment sets alternate by phase parity");
        constexpr int NMP = CJ * PI;                                  // MFMAs per phase
        constexpr int NM = NP * NMP;
        constexpr int EVERY = ILV ? (NM / PER_STAGE > 0 ? NM / PER_STAGE : 1) : 1;
        // pieces that ride on the MFMAs of phases 0 .. NP-2 (before the wait); the last phase carries the rest, after it
        constexpr int PB_RAW = ILV ? ((NP - 1) * NMP + EVERY - 1) / EVERY : PER_STAGE;
        constexpr int PB = PB_RAW < PER_STAGE ? PB_RAW : PER_STAGE;
        static_assert(!ILV || STAGES >= 3, "interleaved issue needs the tile after next in flight");
        auto read_frags = [&](int b, int s_, uint4 (&wf)[CJ], uint4 (&xf)[PI]) {
            const unsigned char* xs = smem + b * S::STAGE + (wp * (PTL / 2)) * KB + row_off;
            const unsigned char* ws = smem + b * S::STAGE + S::XB + (wc * (CT / 2)) * KB + row_off;
            const int slot = ((s_ * 2 + khalf) ^ sw) * 16;
#pragma unroll
            for (int j = 0; j < CJ; ++j) wf[j] = *(const uint4*)(ws + j * 32 * KB + slot);
#pragma unroll
            for (int i = 0; i < PI; ++i) xf[i] = *(const uint4*)(xs + i * 32 * KB + slot);
        };
        uint4 wfa[CJ], xfa[PI], wfb[CJ], xfb[PI];
        read_frags(0, 0, wfa, xfa);
        for (int kt = kt0; kt < kt1; ++kt) {
            issue_tile(kt + STAGES - 1, nbuf);
            const int cur_nbuf = nbuf;
#pragma unroll
            for (int s_ = 0; s_ < NP; ++s_) {
                uint4 (&wfc)[CJ] = (s_ & 1) ? wfb : wfa;
                uint4 (&xfc)[PI] = (s_ & 1) ? xfb : xfa;
                uint4 (&wfn)[CJ] = (s_ & 1) ? wfa : wfb;
                uint4 (&xfn)[PI] = (s_ & 1) ? xfa : xfb;
                if (s_ + 1 < NP) {
                    read_frags(buf, s_ + 1, wfn, xfn);
                } else {
                    // tile kt+1 has landed (this wave's share; the barrier extends it to all waves)
                    wait_vmcnt<(STAGES - 3 >= 0 ? STAGES - 3 : 0) * PER_STAGE + (STAGES >= 3 ? PB : 0)>();
                    __syncthreads();
                    buf = buf + 1 == STAGES ? 0 : buf + 1;
                    nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
                    if (kt + 1 < kt1) read_frags(buf, 0, wfn, xfn);
                }
                mfma_prio<1>();
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < PI; ++i) {
                        Mma<T>::run(wfc[j], xfc[i], acc[j][i]);
                        if constexpr (ILV) {
                            const int m = (s_ * CJ + j) * PI + i;
                            if (m % EVERY == 0 && m / EVERY < PER_STAGE) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(m / EVERY, cur_nbuf);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                mfma_prio<0>();
            }
            if constexpr (ILV) {   // pieces the MFMA count of a tile could not carry
                constexpr int DONE = (NM + EVERY - 1) / EVERY < PER_STAGE ? (NM + EVERY - 1) / EVERY : PER_STAGE;
#pragma unroll
                for (int pc = DONE; pc < PER_STAGE; ++pc) issue_piece(pc, cur_nbuf);
            }
        }
        wait_vmcnt<0>();
        __syncthreads();
        if (IG_ABL(128)) return;
        conv_epilogue<T, CT, S::CRS, MODE, PTL, (MINW >= 4 ? 2 : 4)>(p, smem, acc, tile, p0, c0, tid, lane, wave, wp, wc);
        return;
    }
    for (int kt = kt0; kt < kt1; ++kt) {
        if (!IG_ABL(1)) issue_tile(kt + STAGES - 1, nbuf);
        const unsigned char* xs = smem + buf * S::STAGE + (wp * (PTL / 2)) * KB + row_off;
        const unsigned char* ws = smem + buf * S::STAGE + S::XB + (wc * (CT / 2)) * KB + row_off;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const int slot = ((s * 2 + khalf) ^ sw) * 16;
            uint4 wf[CJ], xf[PI];
#pragma unroll
            for (int j = 0; j < CJ; ++j) wf[j] = *(const uint4*)(ws + j * 32 * KB + slot);
#pragma unroll
            for (int i = 0; i < PI; ++i) xf[i] = *(const uint4*)(xs + i * 32 * KB + slot);
            if (IG_ABL(2)) {   // keep the LDS reads, drop the matrix work
#pragma unroll
                for (int j = 0; j < CJ; ++j) asm volatile("" ::"v"(wf[j].x), "v"(wf[j].w));
#pragma unroll
                for (int i = 0; i < PI; ++i) asm volatile("" ::"v"(xf[i].x), "v"(xf[i].w));
            } else {
                mfma_prio<1>();
                if constexpr (ILV) {
                    constexpr int NM = (KC / 2) * CJ * PI;                   // MFMA groups per K tile
                    constexpr int EVERY = NM / PER_STAGE > 0 ? NM / PER_STAGE : 1;
#pragma unroll
                    for (int j = 0; j < CJ; ++j)
#pragma unroll
                        for (int i = 0; i < PI; ++i) {
                            Mma<T>::run(wf[j], xf[i], acc[j][i]);
                            const int m = (s * CJ + j) * PI + i;
                            if (!IG_ABL(1) && m % EVERY == 0 && m / EVERY < PER_STAGE) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(m / EVERY, nbuf);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < CJ; ++j)
#pragma unroll
                        for (int i = 0; i < PI; ++i) Mma<T>::run(wf[j], xf[i], acc[j][i]);
                }
                mfma_prio<0>();
            }
        }
        if constexpr (ILV) {   // pieces the MFMA count of a tile could not carry
            constexpr int NM = (KC / 2) * CJ * PI;
            constexpr int EVERY = NM / PER_STAGE > 0 ? NM / PER_STAGE : 1;
            constexpr int DONE = (NM + EVERY - 1) / EVERY < PER_STAGE ? (NM + EVERY - 1) / EVERY : PER_STAGE;
#pragma unroll
            for (int pc = DONE; pc < PER_STAGE; ++pc)
                if (!IG_ABL(1)) issue_piece(pc, nbuf);
        }
        // tile kt+1 must have landed (this wave's share; the barrier extends it to all waves); the STAGES-2 younger
        // tiles stay in flight across the barrier
        wait_vmcnt<(STAGES - 2) * PER_STAGE>();
        if (!IG_ABL(4)) __syncthreads();
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (IG_ABL(128)) {   // no epilogue (one dummy store keeps the accumulators alive)
        float t = 0.f;
        for (int j = 0; j < CJ; ++j) for (int i = 0; i < PI; ++i) t += acc[j][i][0];
        if (t == 1.2345f) ((float*)p.out)[0] = t;
        return;
    }
    // rows in flight per thread in the epilogue: the 128-VGPR (4 workgroups/CU) configuration has no room for more than 2
    conv_epilogue<T, CT, S::CRS, MODE, PTL, (MINW >= 4 ? 2 : 4)>(p, smem, acc, tile, p0, c0, tid, lane, wave, wp, wc);
}

__global__ void relu_inplace_kernel(float* x, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ((float4*)x)[i];
        ((float4*)x)[i] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
}

template <typename T, int CT, int MODE>
int launch(ConvParams& p, hipStream_t stream) {
    constexpr bool BWD = MODE != 0;   // (anything but the lean forward epilogue)
    static int dlds_min_k = vince_knob("dlds_min_k", 0);
    const int k_elems = p.total_chunks * (16 / (int)sizeof(T));
    // VINCE_DLDS_CFG=4 forces the 128-pixel tile everywhere (measurement aid); the default (5) adds the 256-pixel tile
    static int dlds_cfg = VINCE_MEASURE_KNOB("dlds_cfg", 5);
    static int big_min_k = vince_knob("big_min_k", 1024);
    static int big_min_tiles = vince_knob("big_min_tiles", 256);
    static long narrow256 = vince_knob("narrow256_min_tiles", 2048);   // 0 = off
    // rotated main loop (fragment reads one MFMA phase ahead, across the tile barrier): bit 0 the 256x128 tile, 1 the 256x64
    // tile, 2 the 2-stage 128-pixel tile, 3 the 3-stage 128-pixel tile
    static int rot = vince_knob("rot", 9);
    static int rot_min_k = VINCE_MEASURE_KNOB("rot_min_k", 0);
    if (p.in_bytes && p.w_bytes && k_elems >= dlds_min_k) {
        const int cpt = p.cpt_mask == 0x7fffffff ? p.total_chunks : p.cpt_mask + 1;
        p.uniform_taps = (cpt % 4 == 0) && (p.total_chunks % 4 == 0);
        p.nkt = (p.total_chunks + 3) / 4;     // 64-byte K rows
        if (p.in2 && !p.uniform_taps) {
            vince_set_error("vince_conv_igemm: in2 needs Ci and in2_channels to be multiples of a 64-byte K row");
            return VINCE_E_UNSUPPORTED;
        }
        // 1x1 reductions at least VINCE_KC8_MIN_K long on the 128-pixel tile: 128-byte K rows (KC = 8), i.e. whole cache lines per
        // DMA'd row -- 64-byte row pieces are request-bound (tools/micro/feed_micro: 14 B/clk/CU against 31 with whole lines);
        // two stages of 32 KB, two workgroups per CU.  Measured: 2048 -> 512 at 7x7 43.6 -> 37.8 us; shorter reductions and the
        // 3x3 layers lose more to the halved occupancy than the whole lines give back.
        static int kc8_min_k = vince_knob("kc8_min_k", 2048);   // 0 = off
        if constexpr (sizeof(T) == 2 && CT == 128) {
            const bool big = dlds_cfg == 5 && k_elems >= big_min_k && (long)((p.M + 255) / 256) * p.ctiles >= big_min_tiles;
            if (kc8_min_k > 0 && !big && p.cpt_mask == 0x7fffffff && p.total_chunks % 8 == 0 && k_elems >= kc8_min_k) {
                p.uniform_taps = 1;
                p.nkt = p.total_chunks / 8;
                hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 128, 8, 2, 2, PT, MODE>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                VINCE_CHECK_LAUNCH();
                return VINCE_OK;
            }
        }
        if (dlds_cfg == 5 && CT == 128 && k_elems >= big_min_k && (long)((p.M + 255) / 256) * p.ctiles >= big_min_tiles) {
            // 256-pixel tiles, 3 stages (2 workgroups per CU): long reductions with enough tiles to fill the chip --
            // 25 % fewer operand bytes per FLOP through the L2 -> LDS path that bounds the 128-pixel tile
            if constexpr (CT == 128) {
                p.ptiles = (p.M + 255) / 256;
                p.variant = 1;
                if (rot & 1) {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 128, 4, 3, 2, 256, MODE, true>), dim3(p.ptiles * p.ctiles), dim3(256), 0,
                                       stream, p);
                } else {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 128, 4, 3, 2, 256, MODE>), dim3(p.ptiles * p.ctiles), dim3(256), 0,
                                       stream, p);
                }
            }
        } else if (CT == 64 && narrow256 && (long)((p.M + 255) / 256) >= narrow256) {
            // 64-channel layers with very many pixel tiles (stem, layer1): 256-pixel tiles halve the per-tile fixed cost
            // (DMA latency, LDS transpose, statistics) and move 17 % fewer operand bytes per FLOP
            if constexpr (CT == 64) {
                p.ptiles = (p.M + 255) / 256;
                p.variant = 1;
                if (rot & 2) {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 64, 4, 2, 3, 256, MODE, true>), dim3(p.ptiles * p.ctiles), dim3(256), 0,
                                       stream, p);
                } else {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 64, 4, 2, 3, 256, MODE>), dim3(p.ptiles * p.ctiles), dim3(256), 0,
                                       stream, p);
                }
            }
        } else {   // 128-pixel tiles, 2 stages, registers capped for 4 workgroups per CU
            // Tiny-M fp32 GEMMs (the projection MLP, 256 rows: 2 pixel tiles) would leave most CUs idle: split the
            // reduction over grid.y, partial sums meet in a zeroed output through fp32 atomics, ReLU runs afterwards.
            int splits = 1;
            const long tiles = (long)p.ptiles * p.ctiles;
            static const bool splitk_env = (VINCE_MEASURE_KNOB("splitk", 1) != 0);
            if (sizeof(T) == 4 && splitk_env && !BWD && !p.e.stats && tiles < 128 && p.nkt >= 16 &&
                p.d.osh == 1 && p.d.osw == 1 && p.d.OH == p.d.Ho && p.d.OW == p.d.Wo) {
                static const long target = VINCE_MEASURE_KNOB("splitk_wgs", 256);   // (env: measurement aid) more splits cost more in atomics than they buy
                splits = (int)min((long)(p.nkt / 8), (target + tiles - 1) / tiles);
                if (splits < 2) splits = 1;
            }
            if (splits > 1) {
                const int relu = p.e.flags & VINCE_EPI_RELU;
                p.e.flags &= ~VINCE_EPI_RELU;
                p.kt_per_split = (p.nkt + splits - 1) / splits;
                splits = (p.nkt + p.kt_per_split - 1) / p.kt_per_split;
                const size_t n = (size_t)p.M * p.d.Co;
                if (int zrc = vince_zero_async(p.out, n * sizeof(float), stream)) return zrc;
                hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 2, 4, PT, MODE>), dim3(p.ptiles * p.ctiles, splits), dim3(256), 0,
                                   stream, p);
                if (relu) hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)min((size_t)1024, (n / 4 + 255) / 256)), dim3(256), 0,
                                             stream, (float*)p.out, n / 4);
            } else {
                // reductions at least VINCE_S3_MIN_K long take a 3-stage ring (two K tiles in flight, 3 workgroups per CU) instead
                // of 2 stages / 4 workgroups
                // default 2048: layer4's 3x3 (K = 4608) 92.6 -> 85 us, 2048 -> 512 50 -> 44 us; shorter reductions lose
                static const int s3_min_k = vince_knob("s3_min_k", 2048);
                if (s3_min_k > 0 && k_elems >= s3_min_k) {
                    if (rot & 8)
                        hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 3, 3, PT, MODE, true>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                    else
                        hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 3, 3, PT, MODE>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                } else if ((rot & 4) && k_elems >= rot_min_k) {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 2, 4, PT, MODE, true>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                } else {
                    hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 2, 4, PT, MODE>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                }
            }
        }
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    if (p.in2) {
        vince_set_error("vince_conv_igemm: in2 needs the direct-to-LDS kernels (tensors < 2 GiB, taps a whole number of K tiles)");
        return VINCE_E_UNSUPPORTED;
    }
    // register-staged fallback (tensors beyond the 31-bit buffer offsets of the direct-to-LDS path): K tile = 128 bytes
    // per row (8 chunks) when the reduction is long enough to pipeline, else 64 bytes.  Generic epilogue.
    p.variant = 2;
    if (k_elems >= 1024) {
        p.nkt = (p.total_chunks + 7) / 8;
        hipLaunchKernelGGL((conv_igemm_kernel<T, CT, 8, (MODE == 2 ? 2 : 1)>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
    } else {
        p.nkt = (p.total_chunks + 3) / 4;
        hipLaunchKernelGGL((conv_igemm_kernel<T, CT, 4, (MODE == 2 ? 2 : 1)>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
    }
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

}  // namespace

extern "C" int vince_conv_igemm(const vince_conv_desc* dd, int dtype, const void* in, const void* w, void* out,
                                const vince_conv_epi* epi, void* stream) {
    vince_conv_epi e;
    if (epi) e = *epi; else memset(&e, 0, sizeof(e));
    VINCE_CHECK_ARG(dd && in && w && out, VINCE_E_ARG, "vince_conv_igemm: null pointer");
    if (e.replicas <= 0 || e.replicas > VINCE_STATS_REPLICAS) e.replicas = VINCE_STATS_REPLICAS;
    VINCE_CHECK_ARG(!e.acc_mask || (e.flags & VINCE_EPI_ACCUMULATE), VINCE_E_ARG, "vince_conv_igemm: acc_mask needs VINCE_EPI_ACCUMULATE");
    VINCE_CHECK_ARG(!e.bnred.y || (e.bnred.mean && e.bnred.invstd && e.bnred.sums && !e.stats), VINCE_E_ARG,
                    "vince_conv_igemm: bnred needs y, mean, invstd and sums, and excludes stats");
    VINCE_CHECK_ARG(!e.out_mask || (!e.out_scale && !e.id_scale && !e.bnred.y), VINCE_E_ARG,
                    "vince_conv_igemm: out_mask belongs to the gradient epilogue and excludes bnred / the forward join");
    VINCE_CHECK_ARG(!e.id_scale == !e.id_shift, VINCE_E_ARG, "vince_conv_igemm: id_scale and id_shift come together");
    VINCE_CHECK_ARG((!e.out_scale && !e.id_scale) || ((e.flags & VINCE_EPI_ACCUMULATE) && !e.acc_mask && !e.bnred.y && !e.stats), VINCE_E_ARG,
                    "vince_conv_igemm: out_scale / id_scale need VINCE_EPI_ACCUMULATE and exclude acc_mask, bnred and stats");
    VINCE_CHECK_ARG(!e.bnred.mask_scale == !e.bnred.mask_shift, VINCE_E_ARG,
                    "vince_conv_igemm: bnred mask_scale and mask_shift come together");
    VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv_igemm: bad dtype %d", dtype);
    VINCE_CHECK_ARG(!e.in2 || (dd->TA * dd->TB >= 2 && dd->Cs == 0 && e.in2_channels > 0 && e.in2_channels <= dd->Ci &&
                               e.in2_channels % (dtype == VINCE_F32 ? 4 : 8) == 0 && ((uintptr_t)e.in2 & 15) == 0), VINCE_E_ARG,
                    "vince_conv_igemm: in2 is the input of the last of at least two taps, in2_channels <= Ci, 16-byte aligned");
    const vince_conv_desc& d = *dd;
    const int CH = dtype == VINCE_F32 ? 4 : 8;
    VINCE_CHECK_ARG(d.N > 0 && d.Hi > 0 && d.Wi > 0 && d.Ho > 0 && d.Wo > 0 && d.Co > 0 && d.Ci > 0, VINCE_E_SHAPE,
                    "vince_conv_igemm: non-positive dimension");
    VINCE_CHECK_ARG(d.Ci % CH == 0, VINCE_E_SHAPE, "vince_conv_igemm: Ci=%d not a multiple of %d", d.Ci, CH);
    if (d.Cs > 0) {   // packed row taps: every tap start must stay 16-byte aligned and inside its input row
        const int eb = dtype == VINCE_F32 ? 4 : 2;
        VINCE_CHECK_ARG(d.TB == 1 && d.Cs < d.Ci && d.Ci % d.Cs == 0 && d.Kw > 0 && d.Kw <= d.Ci / d.Cs, VINCE_E_SHAPE,
                        "vince_conv_igemm: packed row taps need TB=1, Cs | Ci, 0 < Kw <= Ci/Cs");
        VINCE_CHECK_ARG((d.Cs * eb) % 8 == 0 && (d.sw * d.Cs * eb) % 16 == 0 && (d.dw0 * d.Cs * eb) % 16 == 0 &&
                        (d.Wi * d.Cs * eb) % 16 == 0, VINCE_E_ALIGN, "vince_conv_igemm: packed row taps must start 16-byte aligned");
        VINCE_CHECK_ARG(d.dw0 >= 0 && (d.Wo - 1) * d.sw + d.dw0 + d.Ci / d.Cs <= d.Wi, VINCE_E_SHAPE,
                        "vince_conv_igemm: packed row taps must stay inside the (padded) input row");
    }
    VINCE_CHECK_ARG(d.Co % CH == 0, VINCE_E_SHAPE, "vince_conv_igemm: Co=%d not a multiple of %d", d.Co, CH);
    VINCE_CHECK_ARG(d.TA >= 1 && d.TB >= 1 && d.TB <= 8 && d.TA * d.TB <= 64, VINCE_E_SHAPE,
                    "vince_conv_igemm: tap grid %dx%d unsupported", d.TA, d.TB);
    VINCE_CHECK_ARG((long long)d.N * d.Ho * d.Wo < (1ll << 31), VINCE_E_SHAPE, "vince_conv_igemm: too many output pixels");
    VINCE_CHECK_ARG((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN,
                    "vince_conv_igemm: pointers must be 16-byte aligned");
    ConvParams p;
    p.d = d;
    const int T = d.TA * d.TB, cpt = d.Ci / CH;
    if (T == 1) {
        p.log2_cpt = 31;
        p.cpt_mask = 0x7fffffff;
    } else {
        VINCE_CHECK_ARG((cpt & (cpt - 1)) == 0, VINCE_E_SHAPE,
                        "vince_conv_igemm: Ci/%d = %d must be a power of two for multi-tap convs", CH, cpt);
        int l = 0;
        while ((1 << l) < cpt) ++l;
        p.log2_cpt = l;
        p.cpt_mask = cpt - 1;
    }
    p.total_chunks = T * cpt;
    if (e.in2) p.total_chunks = (T - 1) * cpt + e.in2_channels / CH;   // the last tap reads the (narrower) second tensor
    p.nkt = (p.total_chunks + 3) / 4;
    p.M = d.N * d.Ho * d.Wo;
    p.tb_mul = (65536 + d.TB - 1) / d.TB;
    p.div_howo = make_fastdiv((uint32_t)(d.Ho * d.Wo));
    p.div_wo = make_fastdiv((uint32_t)d.Wo);
    p.in = in; p.w = w; p.out = out; p.e = e;
    p.variant = 0;
    p.cs = d.Cs > 0 ? d.Cs : d.Ci;
    p.in2 = e.in2;
    p.cs2 = e.in2_channels;
    p.in2_bytes = 0;
    p.kt_per_split = 0;
    p.ablate = 0;
#ifdef VINCE_MEASURE
    static int ablate = VINCE_MEASURE_KNOB("conv_ablate", 0);
    p.ablate = ablate;
#endif
    {
        const unsigned long long esz = dtype == VINCE_F32 ? 4 : 2;
        const unsigned long long ib = (unsigned long long)d.N * d.Hi * d.Wi * p.cs * esz, wb = (unsigned long long)d.Co * d.WT * d.Ci * esz;
        p.in_bytes = ib < 0x7ff00000ull ? (uint32_t)ib : 0;   // the direct-to-LDS path addresses with 31-bit offsets
        if (e.in2) {
            const unsigned long long ib2 = (unsigned long long)d.N * d.Hi * d.Wi * e.in2_channels * esz;
            p.in2_bytes = ib2 < 0x7ff00000ull ? (uint32_t)ib2 : 0;
            if (!p.in2_bytes) p.in_bytes = 0;
        }
        p.w_bytes = wb < 0x7ff00000ull ? (uint32_t)wb : 0;
    }
    p.ptiles = (p.M + PT - 1) / PT;
    // 64-channel tiles for short reductions: such layers are HBM-bound and the smaller accumulator footprint buys
    // occupancy (5 waves/SIMD vs 3), which is what a streaming kernel needs
    static int ct64_max_k = VINCE_MEASURE_KNOB("ct64_max_k", 0);
    bool narrow = d.Co <= 64 || (d.TA * d.TB * d.Ci <= ct64_max_k);
    // tiny-M GEMMs (the projection MLP: 256 rows -> 2 pixel tiles): 64-channel tiles double the workgroup count
    if (!narrow && (long)p.ptiles * ((d.Co + 127) / 128) < 128) narrow = true;
    const int CT = narrow ? 64 : 128;
    p.ctiles = (d.Co + CT - 1) / CT;
    hipStream_t s = (hipStream_t)stream;
    void* tok = nullptr;
    if (vince_profile_enabled()) {
        // algorithmic FLOPs: the stem's input channels are padded 3 -> CH; count the 3 real ones
        // (the stem: 3 real channels behind the padding, and Kw real pixels per packed row tap)
        const double ci_alg = d.Cs > 0 ? 3.0 * d.Kw : ((d.Ci == CH && T > 1) ? 3.0 : (double)d.Ci);
        vince_profile_begin_launch(0, 2.0 * p.M * d.Co * T * ci_alg, stream, &tok);
        vince_profile_set_dims(tok, p.M, d.Co, T * d.Ci, T, d.sh * 10 + d.osh, e.flags);
    }
    int rc;
    const bool bwd = (e.flags & VINCE_EPI_ACCUMULATE) || e.bnred.y || e.out_mask;   // gradient epilogue instantiation
    const bool join = e.out_scale || e.id_scale;   // forward residual join with known BatchNorm constants
    // The 8-wavefront 256 x 256 core (conv_m8.hip) takes the long bf16 reductions with 256-channel output tiles: the 3x3 and wide
    // 1x1 convolutions of layer3 / layer4 and their input gradients.  VINCE_M8=0 keeps everything on this file's tiles (the
    // cross-check of the parity tests), VINCE_M8_MIN_K moves the threshold.
    static const int m8_on = vince_knob("m8", 1);
    static const int m8_min_k = vince_knob("m8_min_k", 1024);
    static const int m8_min_tiles = vince_knob("m8_min_tiles", 128);
    rc = -1;
    if (m8_on && !e.in2 && dtype == VINCE_BF16 && d.Co % 256 == 0 && T * d.Ci >= m8_min_k &&
        (long)((p.M + 255) / 256) * (d.Co / 256) >= m8_min_tiles)
        rc = vince_conv_m8_launch(p, join ? 2 : (bwd ? 1 : 0), s);
    if (rc != -1) {
        if (tok) {
            vince_profile_set_tag(tok, VINCE_TAG_M8_FWD + (bwd ? 1 : 0));
            vince_profile_end_launch(tok, stream);
        }
        return rc;
    }
    if (dtype == VINCE_F32) {
        if (join) rc = narrow ? launch<float, 64, 2>(p, s) : launch<float, 128, 2>(p, s);
        else if (bwd) rc = narrow ? launch<float, 64, 1>(p, s) : launch<float, 128, 1>(p, s);
        else rc = narrow ? launch<float, 64, 0>(p, s) : launch<float, 128, 0>(p, s);
    } else {
        if (join) rc = narrow ? launch<bf16_t, 64, 2>(p, s) : launch<bf16_t, 128, 2>(p, s);
        else if (bwd) rc = narrow ? launch<bf16_t, 64, 1>(p, s) : launch<bf16_t, 128, 1>(p, s);
        else rc = narrow ? launch<bf16_t, 64, 0>(p, s) : launch<bf16_t, 128, 0>(p, s);
    }
    if (tok) {
        // one tag per kernel symbol: [dtype][64ch | 128ch] x [128px | 256px][fwd | bwd epilogue]; the register-staged
        // fallback (never taken at the benchmark sizes) is counted with the 128-pixel tile of its shape
        const int shape = (narrow ? 0 : 2) + (p.variant == 1 ? 1 : 0);   // 64ch x 128px, 64ch x 256px, 128ch x 128px, 128ch x 256px
        vince_profile_set_tag(tok, (dtype == VINCE_F32 ? 0 : 8) + shape * 2 + (bwd ? 1 : 0));
        vince_profile_end_launch(tok, stream);
    }
    return rc;
}
