// The implicit-GEMM convolution kernels and their tile selection (launch<T, CT, MODE>), shared by the translation units that
// instantiate them: conv_igemm.hip (float, bf16) and conv_igemm_x3.hip (the split-half element types x3h_t / x3b_t).
#pragma once
#ifndef VINCE_UBM4
#define VINCE_UBM4 2
#endif
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
        if constexpr (X3<T>::on) {   // split-half products: two 16-byte fragments (8 floats) feed one 32x32x16 block
#pragma unroll
            for (int s = 0; s < KC / 2; s += 2) {
                uint4 wh[CJ], wl[CJ], xh[2], xl[2];
#pragma unroll
                for (int j = 0; j < CJ; ++j) {   // weights arrive split (vince_prepare_weight, VINCE_F32X3): chunk h = hi, chunk 2 + h = lo
                    wh[j] = *(const uint4*)(ws + j * 32 * RS + s * 32);
                    wl[j] = *(const uint4*)(ws + j * 32 * RS + s * 32 + 32);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    x3_split<T, false>(*(const uint4*)(xs + i * 32 * RS + s * 32), *(const uint4*)(xs + i * 32 * RS + s * 32 + 32), xh[i], xl[i]);
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) x3_mma<T>(wh[j], wl[j], xh[i], xl[i], acc[j][i]);
            }
        } else {
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
        }
        if (kt + 1 < p.nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    x3_unscale<T>(acc);
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
    // the split-half types leave through the epilogue in two 64-channel passes (ECT): an fp32 tile of 128 x 128 outputs would hold the
    // kernel to two workgroups per CU by its LDS alone
    static constexpr int ECT = (X3<T>::on && CT == 128) ? 64 : CT;
    static constexpr int CRS = ECT * (int)sizeof(T) + 16;
    static constexpr int EPI = PTL * CRS + 4 * ECT * 2 * 4;
    static constexpr int BYTES0 = MAIN > EPI ? MAIN : EPI;
};

// K tile = KC 16-byte chunks per row; STAGES-deep LDS ring, prefetch distance STAGES-1 tiles, counted vmcnt so that the
// younger tiles stay in flight across the barrier (one barrier per K tile).
// PTL = pixels per workgroup tile (128 or 256).  The L2 -> LDS fill rate of a CU (measured ~19 B/clk with every CU
// streaming) caps a 128x128 tile at ~700 TFLOP/s chip-wide: 256 B of operands per K element feed 32768 FLOP.  The
// 256-pixel tile moves 25 % fewer bytes per FLOP (each wave owns 128 pixels x CT/2 channels).
// WN = wavefronts along the channel axis (the other 4 / WN split the pixels): 2 = the 2 x 2 arrangement, 1 = every wavefront holds all CT
// channels of PTL / 4 pixels.
template <typename T, int CT, int KC, int STAGES, int MINW, int PTL, int MODE, bool ROT = false, int WN = 2>
__global__ __launch_bounds__(256, MINW) void conv_igemm_dlds_kernel(const ConvParams p) {
    constexpr int CH = Elem<T>::CH;
    constexpr int WP = 4 / WN;
    constexpr int CJ = CT / (32 * WN), PI = PTL / (32 * WP);
    static_assert(!ROT || WN == 2, "the rotated loop is written for the 2 x 2 arrangement");
    using S = SmemD<T, CT, KC, STAGES, PTL>;
    constexpr int KB = S::KB;
    constexpr int RPW = 1024 / KB;                 // rows per wave DMA instruction (8 or 16)
    constexpr int RPP = 4 * RPW;                   // rows per pass of the 4 waves
    constexpr int XROWS = PTL / RPP, WROWS = CT / RPP;
    constexpr int PER_STAGE = XROWS + WROWS;       // DMA instructions per thread per stage
    constexpr int SWSH = KC == 8 ? 1 : 2, SWMASK = KC - 1;   // slot swizzle = (row >> SWSH) & SWMASK
    constexpr bool ILV = CT == 128 && STAGES == 3 && !X3<T>::on;    // DMA issue interleaved with the MFMAs (see issue_piece)
    static_assert(!(ROT && X3<T>::on), "the split-half element types take the plain main loop");
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::BYTES0];

    if (IG_ABL(64)) return;   // launch + workgroup dispatch only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WN, wp = wave / WN;
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
        const int ccx = src2 ? (cc & p.cpt2_mask) : cc;        // (in2_repeat: the second tensor's row is read again from its start)
#pragma unroll
        for (int e = 0; e < XROWS; ++e) {
            const int hi = hb[e] + dh, wi = wb[e] + dw;
            const bool ok = rv[e] && qv && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
            offx[e] = ok ? ((nb[e] + (uint32_t)(hi * d.Wi + wi)) * cs + (uint32_t)ccx * CH) * (uint32_t)sizeof(T) : OOB;
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
            // (log2_tapid = log2_cpt, except with in2_repeat: the chunks of one in2 row -- every pass over the second tensor's row then starts
            // like a new tap, the offsets made afresh instead of advanced; the other taps just recompute theirs a few times more)
            const int tap = kt >= p.nkt ? 0x7fffff : ((kt * KC) >> p.log2_tapid);   // wave-uniform
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
        const unsigned char* xs = smem + buf * S::STAGE + (wp * (PTL / WP)) * KB + row_off;
        const unsigned char* ws = smem + buf * S::STAGE + S::XB + (wc * (CT / WN)) * KB + row_off;
        if constexpr (X3<T>::on) {
            // split-half products (x3h_t / x3b_t): the activation fragments of two MFMA phases (8 floats per lane and row) are split
            // into hi / lo halves in registers, the weight fragments come split from the weight cache; three half-precision
            // 32x32x16 MFMAs per output block.  The split is VALU work per ACTIVATION fragment, so these types run with WN = 1: a
            // wavefront owns 32 pixels x all CT channels (one activation fragment per CT / 32 blocks)
#pragma unroll
            for (int s = 0; s < KC / 2; s += 2) {
                const int slot0 = ((s * 2 + khalf) ^ sw) * 16, slot1 = ((s * 2 + 2 + khalf) ^ sw) * 16;
                uint4 wh[CJ], wl[CJ], xh[PI], xl[PI];
#pragma unroll
                for (int j = 0; j < CJ; ++j) {   // weights arrive split (vince_prepare_weight, VINCE_F32X3): chunk h = hi, chunk 2 + h = lo
                    wh[j] = *(const uint4*)(ws + j * 32 * KB + slot0);
                    wl[j] = *(const uint4*)(ws + j * 32 * KB + slot1);
                }
#pragma unroll
                for (int i = 0; i < PI; ++i) {
                    if (p.presplit) {   // (uniform) stored half pairs: chunk h = hi, chunk 2 + h = lo of this lane's eight K elements, as the weights
                        xh[i] = *(const uint4*)(xs + i * 32 * KB + slot0);
                        xl[i] = *(const uint4*)(xs + i * 32 * KB + slot1);
                    } else
                    x3_split<T, false>(*(const uint4*)(xs + i * 32 * KB + slot0), *(const uint4*)(xs + i * 32 * KB + slot1), xh[i], xl[i]);
                }
                mfma_prio<1>();
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < PI; ++i) x3_mma<T>(wh[j], wl[j], xh[i], xl[i], acc[j][i]);
                mfma_prio<0>();
            }
        } else
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
    x3_unscale<T>(acc);
    if constexpr (S::ECT != CT) {
        // two passes of CT / 2 channels each through the (half-size) epilogue tile; WN = 1: a wavefront's accumulators split by index
        static_assert(WN == 1 && CT == 2 * S::ECT, "the two-pass epilogue is written for WN = 1");
        constexpr int HJ = CJ / 2;
        conv_epilogue<T, S::ECT, S::CRS, MODE, PTL, 4, 256, WN>(p, smem, *(f32x16_t (*)[HJ][PI])&acc[0], tile, p0, c0, tid, lane, wave, wp, wc);
        __syncthreads();
        conv_epilogue<T, S::ECT, S::CRS, MODE, PTL, 4, 256, WN>(p, smem, *(f32x16_t (*)[HJ][PI])&acc[HJ], tile, p0, c0 + S::ECT, tid, lane, wave, wp, wc);
        return;
    }
    // rows in flight per thread in the epilogue: the 128-VGPR (4 workgroups/CU) configuration has no room for more than 2
    // (the forward residual join, MODE 2, reads the identity chunk of every row it stores: all of a thread's rows in flight at once --
    // its accumulators are dead by then, so the registers are there; two at a time left four exposed round trips per tile)
    conv_epilogue<T, CT, S::CRS, MODE, PTL, (MODE == 2 ? 8 : MINW >= 4 ? VINCE_UBM4 : 4), 256, WN>(p, smem, acc, tile, p0, c0, tid, lane, wave, wp, wc);
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
    static_assert(!X3<T>::on, "the split-half element types have their own tile selection: launch_x3");
    static int dlds_min_k = vince_knob("dlds_min_k", 0);
    const int k_elems = p.total_chunks * (16 / (int)sizeof(T));
    // VINCE_DLDS_CFG=4 forces the 128-pixel tile everywhere (measurement aid); the default (5) adds the 256-pixel tile
    static int dlds_cfg = VINCE_MEASURE_KNOB("dlds_cfg", 5);
    // (16-bit tensors: 1024 until late round 5, profiles/r05_knob_sweep.txt; the fp32 trunk -- the parity mode -- keeps the arrangement its
    // full-size fixtures were measured with)
    static int big_min_k = vince_knob("big_min_k", sizeof(T) == 2 ? 512 : 1024);
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
            if (sizeof(T) == 4 && splitk_env && !BWD && !p.e.stats && !p.e.out2 && tiles < 128 && p.nkt >= 16 &&
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
                static const int s3_min_k = vince_knob("s3_min_k", sizeof(T) == 2 ? 1024 : 2048);    // (16-bit: 2048 until late round 5, measured layer by layer alone; 1024 is better inside the step)
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

// Tile selection of the split-half element types (x3h_t / x3b_t).  128-pixel tiles, every wavefront over all CT channels of 32 pixels
// (WN = 1: the in-register split is paid per ACTIVATION fragment while the weights arrive split, so one activation fragment should
// feed as many MFMA blocks as the tile has channels), plain main loop.  x3_cfg (cross-check / measurement switch): 1 (default) =
// 128-byte K rows (whole cache lines, half the barriers per reduction element), 2 stages; 0 = 64-byte K rows, 2 or 3 stages by
// reduction length (measured on the 12 layer shapes of ResNet-50 at N = 256: forward 2.26 -> 2.07 ms, input gradients 2.23 -> 2.02).
template <typename T, int CT, int MODE>
int launch_x3(ConvParams& p, hipStream_t stream) {
    static_assert(X3<T>::on, "launch_x3 is for x3h_t / x3b_t");
    const int k_elems = p.total_chunks * 4;
    const int cpt = p.cpt_mask == 0x7fffffff ? p.total_chunks : p.cpt_mask + 1;
    p.uniform_taps = (cpt % 4 == 0) && (p.total_chunks % 4 == 0);
    if (!p.uniform_taps || p.in2) {
        vince_set_error("vince_conv_igemm: the split-half types need Ci to be a multiple of 16 (the split weight layout) and take no in2");
        return VINCE_E_SHAPE;
    }
    if (p.presplit && !(X3<T>::half && p.in_bytes && p.w_bytes)) {
        vince_set_error("vince_conv_igemm: VINCE_EPI_IN_HALF_PAIRS is for VINCE_F32X3H launches on the direct-to-LDS kernels (tensors < 2 GiB)");
        return VINCE_E_UNSUPPORTED;
    }
    const dim3 grid(p.ptiles * p.ctiles);
    if (p.in_bytes && p.w_bytes) {
        static const int x3_cfg = vince_knob("x3_cfg", 1);
        static const int x3_s3_min_k = vince_knob("x3_s3_min_k", 2048);
        p.nkt = p.total_chunks / 4;     // 64-byte K rows
        // (reductions of at most 128 elements -- the 1x1 expand convolutions of layer1 / layer2, pure HBM streams -- do better with the
        // 64-byte rows' three to four workgroups per CU: 255 -> 213 us and 161 -> 144 us at N = 256)
        // 256-pixel tiles (a wavefront owns 64 pixels x 128 channels; 64-byte K rows in a 3-stage ring, two workgroups per CU): a quarter
        // fewer operand bytes per MFMA through the L2 -> LDS path that bounds these kernels (~32 B/clk/CU for 128-byte row pieces against
        // the ~84 the matrix pipe could take from two 128 x 128 workgroups).  Pays where several channel tiles share a pixel tile and the
        // reduction is long: layer3's 3x3 220 -> 198 us, its input gradient 210 -> 181; layer2's 3x3 (one channel tile) loses 3 %.
        // x3_big_min_k = 0: off.
        static const int x3_big_min_k = vince_knob("x3_big_min_k", 512);
        if constexpr (CT == 128) {
            if (x3_big_min_k > 0 && k_elems >= x3_big_min_k && p.ctiles >= 2 && (long)((p.M + 255) / 256) * p.ctiles >= 256) {
                p.ptiles = (p.M + 255) / 256;
                p.variant = 1;
                hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, 128, 4, 3, 2, 256, MODE, false, 1>), dim3(p.ptiles * p.ctiles), dim3(256), 0, stream, p);
                VINCE_CHECK_LAUNCH();
                return VINCE_OK;
            }
        }
        if (x3_cfg == 1 && cpt % 8 == 0 && p.total_chunks % 8 == 0 && k_elems > 128) {
            p.nkt = p.total_chunks / 8;
            // Tiny-M GEMMs (the projection MLP: 256 rows = 2 pixel tiles): split the reduction over grid.y as the fp32 path does --
            // partial sums meet in a zeroed output through fp32 atomics, ReLU runs afterwards (round 5: the head through split-half
            // products, 3 x 16-bit MFMAs per product against exact fp32 MFMAs at a sixteenth of that rate)
            const long tiles = (long)p.ptiles * p.ctiles;
            int splits = 1;
            if (MODE == 0 && !p.e.stats && !p.e.out2 && tiles < 128 && p.nkt >= 16 && p.d.osh == 1 && p.d.osw == 1 && p.d.OH == p.d.Ho && p.d.OW == p.d.Wo) {
                splits = (int)min((long)(p.nkt / 8), (256 + tiles - 1) / tiles);
                if (splits < 2) splits = 1;
            }
            if (splits > 1) {
                const int relu = p.e.flags & VINCE_EPI_RELU;
                p.e.flags &= ~VINCE_EPI_RELU;
                p.kt_per_split = (p.nkt + splits - 1) / splits;
                splits = (p.nkt + p.kt_per_split - 1) / p.kt_per_split;
                const size_t n = (size_t)p.M * p.d.Co;
                if (int zrc = vince_zero_async(p.out, n * sizeof(float), stream)) return zrc;
                hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 8, 2, 2, PT, MODE, false, 1>), dim3(p.ptiles * p.ctiles, splits), dim3(256), 0, stream, p);
                if (relu) hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)min((size_t)1024, (n / 4 + 255) / 256)), dim3(256), 0,
                                             stream, (float*)p.out, n / 4);
                VINCE_CHECK_LAUNCH();
                return VINCE_OK;
            }
            hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 8, 2, 2, PT, MODE, false, 1>), grid, dim3(256), 0, stream, p);
        } else if (k_elems >= x3_s3_min_k) {
            hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 3, 3, PT, MODE, false, 1>), grid, dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((conv_igemm_dlds_kernel<T, CT, 4, 2, 3, PT, MODE, false, 1>), grid, dim3(256), 0, stream, p);
        }
    } else {   // tensors beyond the 31-bit buffer offsets of the direct-to-LDS path: the register-staged kernel (2 x 2 wavefronts)
        p.variant = 2;
        p.nkt = p.total_chunks / 4;
        hipLaunchKernelGGL((conv_igemm_kernel<T, CT, 4, (MODE == 2 ? 2 : 1)>), grid, dim3(256), 0, stream, p);
    }
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

}  // namespace
