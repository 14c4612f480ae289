// Fused query x (keys | queue) similarity + InfoNCE loss + metrics, forward and backward.
//
// Replaces vince_model.py:207-242 (torch.bmm / torch.mm logits, positive mask), utils/loss_util.py:7-62
// (similarity_cross_entropy) and vince_model.py:314-342 (get_metrics).  The B x (Bk+K) logit matrix (67 MB at
// B=256, K=65536) is never written: every workgroup holds a 64-row query tile in registers, streams 128-row
// slabs of the key/queue matrix through LDS, forms 16x16 logit tiles on the matrix cores and keeps per-row online-softmax
// state (running max, sum of exp over negatives, max negative cosine) in registers; rows are reduced across
// the 16 lanes that share them with wavefront shuffles.  A second tiny kernel merges the per-part partials and
// produces the loss, the per-positive distances and the four metrics.
//
// Logits (round 5): SPLIT-HALF products instead of exact fp32 MFMAs.  Both operands are unit vectors; every element x becomes
// hi = half(x * 2^8), lo = half(x * 2^8 - hi) (csrc/common.h x3_split: 22 significand bits, lo halves stay normal down to |x| ~ 5e-4,
// absolute steps of 2e-10 below) and a logit is hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation:
// |error| <= ~2^-22 * sum |q_i k_i| <= 2.4e-7 on a value in [-1, 1] -- fp32's own rounding -- at 96 MFMAs of 16 cycles per
// 16 x 128 tile where v_mfma_f32_16x16x4_f32 took 256 of 32.  The queue slab is split ONCE per workgroup on its way into LDS
// (the same 4 bytes per element: [8 hi halves | 8 lo halves] per group of 8).  Training forwards also WRITE the logits
// (B x (Bk + K) floats, 67 MB at B = 256, K = 65536 -- 13 us of HBM time) so that backward reads them back instead of
// recomputing them next to its own dq contraction.
//
// Loss definition (NOT plain softmax cross-entropy): for row i with positives P_i and negatives N_i,
//   dist_ip = -( s_ip - log( exp(s_ip) + sum_{n in N_i} exp(s_in) ) ),   loss = mean over all (i, p)
// i.e. each positive has its own denominator that excludes the other positives (loss_util.py:37-44).
#include "common.h"

namespace {

constexpr int SL = 128;     // key/queue rows per slab
constexpr int RT = 64;      // query rows per workgroup (16 per wave)
constexpr float NEG_BIG = -1e30f;

struct InfoParams {
    vince_infonce_desc d;
    int parts_inb, parts_q, slabs_per_part_q, nslabs_inb, nslabs_q;
    const float* q;
    const float* inb;
    const float* queue;
    float* pos;
    const float* row_max;
    const float* neg_sum;
    const float* grad_scale;
    float* dq;
    float* wmat;
    float* part;   // [3][P][B]
    float* logits; // optional [B][Bk + K]: written by the forward, read by backward (nullptr: backward recomputes them)
};

template <int D>
struct ISmem {
    static constexpr int SRS = D * 4 + 16;         // slab row stride (bytes)
    static constexpr int SLAB = SL * SRS;
    static constexpr int WRS = SL * 4 + 16;        // weight-tile row stride
    static constexpr int WT = 16 * WRS;            // per wave
};

// Stage slab rows [row0, row0+128) of src ([nrows][D] f32) into LDS, zero-filling rows past nrows.
template <int D>
__device__ inline void stage_slab(unsigned char* lds, const float* __restrict__ src, int row0, int nrows, int tid) {
    constexpr int CPR = D / 4;
    for (int c = tid; c < SL * CPR; c += 256) {
        const int r = c / CPR, j = c % CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows) v = *(const uint4*)(src + (size_t)(row0 + r) * D + j * 4);
        *(uint4*)(lds + r * ISmem<D>::SRS + j * 16) = v;
    }
}

// logits tile: acc[ct][r] = q[row (lane>>4)*4 + r] . slab[ct*16 + (lane&15)]
template <int D>
__device__ inline void logits_tile(const unsigned char* slab, const float4 (&qf)[D / 16], f32x4_t (&acc)[8], int lane) {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* base = slab + (lane & 15) * ISmem<D>::SRS + (lane >> 4) * 16;
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const float4 b = *(const float4*)(base + ct * 16 * ISmem<D>::SRS + kk * 64);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kk].x, b.x, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kk].y, b.y, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kk].z, b.z, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kk].w, b.w, acc[ct], 0, 0, 0);
        }
    }
}

// Which source / column range does part `part` cover?
__device__ inline void part_range(const InfoParams& p, int part, bool& is_inb, int& slab0, int& slab1) {
    if (part < p.parts_inb) {
        is_inb = true;
        slab0 = part;
        slab1 = part + 1;
    } else {
        is_inb = false;
        slab0 = (part - p.parts_inb) * p.slabs_per_part_q;
        slab1 = min(slab0 + p.slabs_per_part_q, p.nslabs_q);
    }
}

template <int D>
__device__ inline void load_q_frags(const InfoParams& p, int rowbase, int lane, float4 (&qf)[D / 16]) {
    // A operand of v_mfma_f32_16x16x4_f32: lane holds row (lane&15), k = (lane>>4); we take 4 consecutive k per
    // 16-byte load and feed them to 4 MFMAs (the slab fragment uses the same k assignment).
    const int row = rowbase + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < p.d.B) v = *(const float4*)(p.q + (size_t)row * D + kk * 16 + (lane >> 4) * 4);
        qf[kk] = v;
    }
}

// ---- split-half logits (see the header) ------------------------------------------------------------------------------------
constexpr float LOGIT_UNSCALE = 1.f / (float)(1 << (2 * X3_WSHIFT));

// Stage slab rows [row0, row0+128) as half pairs: the 8 floats of group j of a row become 16 bytes of hi halves + 16 bytes of lo
// halves at byte j*32 of the row (the row keeps its fp32 footprint and stride), zero rows past nrows.
template <int D>
__device__ inline void stage_slab_split(unsigned char* lds, const float* __restrict__ src, int row0, int nrows, int tid) {
    constexpr int GPR = D / 8;
    for (int c = tid; c < SL * GPR; c += 256) {
        const int r = c / GPR, j = c % GPR;
        uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
        if (row0 + r < nrows) {
            const float* sp = src + (size_t)(row0 + r) * D + j * 8;
            f0 = *(const uint4*)sp;
            f1 = *(const uint4*)(sp + 4);
        }
        uint4 hi, lo;
        x3_split<x3h_t, true>(f0, f1, hi, lo);
        *(uint4*)(lds + r * ISmem<D>::SRS + j * 32) = hi;
        *(uint4*)(lds + r * ISmem<D>::SRS + j * 32 + 16) = lo;
    }
}

// A operand of v_mfma_f32_16x16x32_f16: lane holds query row (lane & 15), k = ks*32 + (lane >> 4)*8 .. +8
template <int D>
__device__ inline void load_q_split(const InfoParams& p, int rowbase, int lane, uint4 (&qh)[D / 32], uint4 (&ql)[D / 32]) {
    const int row = rowbase + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
        uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
        if (row < p.d.B) {
            const float* sp = p.q + (size_t)row * D + ks * 32 + (lane >> 4) * 8;
            f0 = *(const uint4*)sp;
            f1 = *(const uint4*)(sp + 4);
        }
        x3_split<x3h_t, true>(f0, f1, qh[ks], ql[ks]);
    }
}

__device__ __forceinline__ f32x4_t mfma_h(const uint4& a, const uint4& b, f32x4_t c) {
    f16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
}

// logits tile from a slab staged by stage_slab_split: acc[ct][r] = q[row (lane>>4)*4 + r] . slab[ct*16 + (lane&15)]
template <int D>
__device__ inline void logits_tile_split(const unsigned char* slab, const uint4 (&qh)[D / 32], const uint4 (&ql)[D / 32], f32x4_t (&acc)[8], int lane) {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* base = slab + (lane & 15) * ISmem<D>::SRS + (lane >> 4) * 32;
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const uint4 bh = *(const uint4*)(base + ct * 16 * ISmem<D>::SRS + ks * 128);
            const uint4 bl = *(const uint4*)(base + ct * 16 * ISmem<D>::SRS + ks * 128 + 16);
            acc[ct] = mfma_h(qh[ks], bl, acc[ct]);     // small terms first
            acc[ct] = mfma_h(ql[ks], bh, acc[ct]);
            acc[ct] = mfma_h(qh[ks], bh, acc[ct]);
        }
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] *= LOGIT_UNSCALE;
}

template <int D>
__global__ __launch_bounds__(256) void infonce_fwd_partial(const InfoParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ISmem<D>::SLAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rowbase = blockIdx.x * RT + wave * 16;
    const int part = blockIdx.y;
    bool is_inb;
    int slab0, slab1;
    part_range(p, part, is_inb, slab0, slab1);
    const float* src = is_inb ? p.inb : p.queue;
    const int nsrc = is_inb ? p.d.Bk : p.d.K;
    const float invT = p.d.inv_temperature;
    const int F = p.d.frames;

    uint4 qh[D / 32], ql[D / 32];
    load_q_split<D>(p, rowbase, lane, qh, ql);
    const int ldl = p.d.Bk + p.d.K, colbase = is_inb ? 0 : p.d.Bk;      // row length / first column of this source in p.logits

    float m[4], s[4], nm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = NEG_BIG; s[r] = 0.f; nm[r] = NEG_BIG; }

    for (int slab = slab0; slab < slab1; ++slab) {
        __syncthreads();
        stage_slab_split<D>(smem, src, slab * SL, nsrc, tid);
        __syncthreads();
        f32x4_t acc[8];
        logits_tile_split<D>(smem, qh, ql, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowg = rowbase + (lane >> 4) * 4 + r;
            float v[8];
            float lmax = NEG_BIG;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int cg = slab * SL + ct * 16 + (lane & 15);
                float raw = acc[ct][r];
                if (p.logits && rowg < p.d.B && cg < nsrc) p.logits[(size_t)rowg * ldl + colbase + cg] = raw;
                bool neg = cg < nsrc;
                if (is_inb && neg) {
                    const bool is_pos = (cg / F) == (rowg / F);
                    if (is_pos) {
                        if (rowg < p.d.B) p.pos[(size_t)rowg * F + (cg % F)] = raw;
                        neg = false;
                    } else if (!p.d.offdiag_neg) {
                        neg = false;
                    }
                }
                v[ct] = neg ? raw : NEG_BIG;
                lmax = fmaxf(lmax, v[ct]);
            }
            nm[r] = fmaxf(nm[r], lmax);
            const float lm = lmax > 0.5f * NEG_BIG ? lmax * invT : NEG_BIG;
            const float mn = fmaxf(m[r], lm);
            float add = 0.f;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) add += v[ct] > 0.5f * NEG_BIG ? __expf(v[ct] * invT - mn) : 0.f;
            s[r] = s[r] * __expf(m[r] - mn) + add;
            m[r] = mn;
        }
    }
    // merge the 16 lanes that share a row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const float m2 = __shfl_xor(m[r], o, 64), s2 = __shfl_xor(s[r], o, 64), n2 = __shfl_xor(nm[r], o, 64);
            const float mn = fmaxf(m[r], m2);
            s[r] = s[r] * __expf(m[r] - mn) + s2 * __expf(m2 - mn);
            m[r] = mn;
            nm[r] = fmaxf(nm[r], n2);
        }
        const int rowg = rowbase + (lane >> 4) * 4 + r;
        if ((lane & 15) == 0 && rowg < p.d.B) {
            const int P = gridDim.y, B = p.d.B;
            p.part[((size_t)0 * P + part) * B + rowg] = m[r];
            p.part[((size_t)1 * P + part) * B + rowg] = s[r];
            p.part[((size_t)2 * P + part) * B + rowg] = nm[r];
        }
    }
}

// One workgroup of 1024 threads: 4 lanes share a row and split its P partial (max, sum) pairs, so the dependent chain per
// lane is P/4 long (P = 130 at B = 256, K = 65536).
constexpr int MERGE_THREADS = 1024, MERGE_LPR = 4;   // lanes per row
__global__ __launch_bounds__(MERGE_THREADS) void infonce_merge(const InfoParams p, int P, float* row_max, float* neg_sum,
                                                               float* dists, float* sweights, float* scalars) {
    __shared__ float red[5][MERGE_THREADS / MERGE_LPR];
    const int B = p.d.B, F = p.d.frames;
    const float invT = p.d.inv_temperature;
    const int sub = threadIdx.x % MERGE_LPR, slot = threadIdx.x / MERGE_LPR;
    constexpr int ROWS_PER_PASS = MERGE_THREADS / MERGE_LPR;
    float a_loss = 0.f, a_sw = 0.f, a_acc = 0.f, a_pos = 0.f, a_nm = 0.f;
    for (int row0 = 0; row0 < B; row0 += ROWS_PER_PASS) {
        const int row = row0 + slot;
        const bool live = row < B;
        float M = NEG_BIG, nmx = NEG_BIG;
        if (live) {
#pragma unroll 8
            for (int j = sub; j < P; j += MERGE_LPR) {      // (unrolled: eight partials' loads in flight instead of a chain of P / 4 round trips)
                M = fmaxf(M, p.part[((size_t)0 * P + j) * B + row]);
                nmx = fmaxf(nmx, p.part[((size_t)2 * P + j) * B + row]);
            }
        }
#pragma unroll
        for (int o = 1; o < MERGE_LPR; o <<= 1) {
            M = fmaxf(M, __shfl_xor(M, o, 64));
            nmx = fmaxf(nmx, __shfl_xor(nmx, o, 64));
        }
        if (live)
            for (int f = 0; f < F; ++f) M = fmaxf(M, p.pos[(size_t)row * F + f] * invT);   // row max over ALL columns
        float S = 0.f;
        if (live) {
#pragma unroll 8
            for (int j = sub; j < P; j += MERGE_LPR)
                S += p.part[((size_t)1 * P + j) * B + row] * __expf(p.part[((size_t)0 * P + j) * B + row] - M);
        }
#pragma unroll
        for (int o = 1; o < MERGE_LPR; o <<= 1) S += __shfl_xor(S, o, 64);
        if (live && sub == 0) {
            row_max[row] = M;
            neg_sum[row] = S;
            for (int f = 0; f < F; ++f) {
                const float raw = p.pos[(size_t)row * F + f];
                const float sp = raw * invT - M;
                const float ls = sp - logf(expf(sp) + S);
                dists[(size_t)row * F + f] = -ls;
                const float sw = expf(ls);
                sweights[(size_t)row * F + f] = sw;
                a_loss += -ls;
                a_sw += sw;
                a_acc += raw > nmx ? 1.f : 0.f;
                a_pos += raw;
            }
            a_nm += nmx;
        }
    }
    if (sub == 0) {
        red[0][slot] = a_loss; red[1][slot] = a_sw; red[2][slot] = a_acc; red[3][slot] = a_pos; red[4][slot] = a_nm;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double t = 0;
        for (int i = 0; i < ROWS_PER_PASS; ++i) t += red[threadIdx.x][i];
        const double denom = threadIdx.x == 4 ? (double)B : (double)B * F;
        scalars[threadIdx.x] = (float)(t / denom);
    }
    if (threadIdx.x >= 5 && threadIdx.x < 8) scalars[threadIdx.x] = 0.f;
}

template <int D>
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const InfoParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ISmem<D>::SLAB + 4 * ISmem<D>::WT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rowbase = blockIdx.x * RT + wave * 16;
    const int part = blockIdx.y;
    bool is_inb;
    int slab0, slab1;
    part_range(p, part, is_inb, slab0, slab1);
    const float* src = is_inb ? p.inb : p.queue;
    const int nsrc = is_inb ? p.d.Bk : p.d.K;
    const float invT = p.d.inv_temperature;
    const int F = p.d.frames, B = p.d.B;
    unsigned char* wt = smem + ISmem<D>::SLAB + wave * ISmem<D>::WT;

    float4 qf[D / 16];
    const bool saved = p.logits != nullptr;        // (uniform) the forward of this step stored its logits: read, do not recompute
    if (!saved) load_q_frags<D>(p, rowbase, lane, qf);
    const int ldl = p.d.Bk + p.d.K, colbase = is_inb ? 0 : p.d.Bk;

    // per-row constants for this lane's 4 rows
    const float gs = p.grad_scale[0] * invT / ((float)B * (float)F);
    float Mr[4], cr[4], Sr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rowg = rowbase + (lane >> 4) * 4 + r;
        Mr[r] = 0.f; cr[r] = 0.f; Sr[r] = 0.f;
        if (rowg < B) {
            Mr[r] = p.row_max[rowg];
            Sr[r] = p.neg_sum[rowg];
            float c = 0.f;
            for (int f = 0; f < F; ++f) c += 1.f / (expf(p.pos[(size_t)rowg * F + f] * invT - Mr[r]) + Sr[r]);
            cr[r] = c;
        }
    }
    f32x4_t dacc[D / 16];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) dacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int slab = slab0; slab < slab1; ++slab) {
        f32x4_t acc[8];
        if (saved) {   // requested before the staging barrier so that the loads fly beside it
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int cg = slab * SL + ct * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowg = rowbase + (lane >> 4) * 4 + r;
                    acc[ct][r] = (cg < nsrc && rowg < B) ? p.logits[(size_t)rowg * ldl + colbase + cg] : 0.f;
                }
            }
        }
        __syncthreads();
        stage_slab<D>(smem, src, slab * SL, nsrc, tid);
        __syncthreads();
        if (!saved) logits_tile<D>(smem, qf, acc, lane);
        // dloss/dlogit (times 1/T) -> wave-private LDS tile [16 rows][128 cols]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = (lane >> 4) * 4 + r, rowg = rowbase + rl;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int cl = ct * 16 + (lane & 15), cg = slab * SL + cl;
                const float raw = acc[ct][r];
                float w = 0.f;
                if (cg < nsrc && rowg < B) {
                    bool neg = true, is_pos = false;
                    if (is_inb) {
                        is_pos = (cg / F) == (rowg / F);
                        neg = !is_pos && p.d.offdiag_neg;
                    }
                    if (is_pos) {
                        w = -gs * Sr[r] / (expf(raw * invT - Mr[r]) + Sr[r]);
                    } else if (neg) {
                        w = gs * cr[r] * __expf(raw * invT - Mr[r]);
                    }
                    if (is_inb && p.wmat) p.wmat[(size_t)rowg * p.d.Bk + cg] = w;
                }
                *(float*)(wt + rl * ISmem<D>::WRS + cl * 4) = w;
            }
        }
        __syncthreads();
        // dq[16 x D] += w[16 x 128] . slab[128 x D]
        const unsigned char* abase = wt + (lane & 15) * ISmem<D>::WRS + (lane >> 4) * 16;
        const unsigned char* bbase = smem + ((lane >> 4) * 4) * ISmem<D>::SRS + (lane & 15) * 4;
#pragma unroll 2
        for (int jj = 0; jj < SL / 16; ++jj) {
            const float4 a = *(const float4*)(abase + jj * 64);
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const unsigned char* bp = bbase + (jj * 16) * ISmem<D>::SRS + dt * 64;
                const float b0 = *(const float*)(bp), b1 = *(const float*)(bp + ISmem<D>::SRS);
                const float b2 = *(const float*)(bp + 2 * ISmem<D>::SRS), b3 = *(const float*)(bp + 3 * ISmem<D>::SRS);
                dacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, dacc[dt], 0, 0, 0);
                dacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, dacc[dt], 0, 0, 0);
                dacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, dacc[dt], 0, 0, 0);
                dacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, dacc[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowg = rowbase + (lane >> 4) * 4 + r;
            if (rowg < B) unsafeAtomicAdd(p.dq + (size_t)rowg * D + dt * 16 + (lane & 15), dacc[dt][r]);
        }
}


// ---- row-wise similarity_cross_entropy on MATERIALISED similarities + arbitrary boolean mask (loss_util.py:7-62,
// equal positives per row).  One wavefront per row; used when a caller hands VinceModel.loss a real tensor.
__device__ inline int lanes_below(unsigned long long ballot, int lane) {
    return __popcll(ballot & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(256) void sce_rows_fwd_kernel(const float* __restrict__ sims, const uint8_t* __restrict__ mask,
                                                           int B, int cols, int P, float invT, float* dists, float* sw,
                                                           float* row_max, float* neg_sum) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* s = sims + (size_t)row * cols;
    const uint8_t* m = mask + (size_t)row * cols;
    float M = NEG_BIG;
    for (int j = lane; j < cols; j += 64) M = fmaxf(M, s[j] * invT);
    M = wave_max(M);
    float S = 0.f;
    for (int j = lane; j < cols; j += 64) S += m[j] ? 0.f : expf(s[j] * invT - M);
    S = wave_sum(S);
    int base = 0;
    for (int j0 = 0; j0 < cols; j0 += 64) {
        const int j = j0 + lane;
        const bool isp = j < cols && m[j];
        const unsigned long long b = __ballot(isp);
        if (isp) {
            const int pi = base + lanes_below(b, lane);
            if (pi < P) {
                const float sp = s[j] * invT - M;
                const float ls = sp - logf(expf(sp) + S);
                dists[(size_t)row * P + pi] = -ls;
                sw[(size_t)row * P + pi] = expf(ls);
            }
        }
        base += __popcll(b);
    }
    if (lane == 0) { row_max[row] = M; neg_sum[row] = S; }
}

__global__ __launch_bounds__(256) void sce_rows_bwd_kernel(const float* __restrict__ sims, const uint8_t* __restrict__ mask,
                                                           int B, int cols, int P, float invT,
                                                           const float* __restrict__ row_max,
                                                           const float* __restrict__ neg_sum,
                                                           const float* __restrict__ gd, float* __restrict__ dsims) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* s = sims + (size_t)row * cols;
    const uint8_t* m = mask + (size_t)row * cols;
    const float M = row_max[row], S = neg_sum[row];
    float c = 0.f;
    int base = 0;
    for (int j0 = 0; j0 < cols; j0 += 64) {
        const int j = j0 + lane;
        const bool isp = j < cols && m[j];
        const unsigned long long b = __ballot(isp);
        if (isp) {
            const int pi = base + lanes_below(b, lane);
            if (pi < P) c += gd[(size_t)row * P + pi] / (expf(s[j] * invT - M) + S);
        }
        base += __popcll(b);
    }
    c = wave_sum(c);
    base = 0;
    for (int j0 = 0; j0 < cols; j0 += 64) {
        const int j = j0 + lane;
        const bool isp = j < cols && m[j];
        const unsigned long long b = __ballot(isp);
        if (j < cols) {
            float g;
            if (isp) {
                const int pi = base + lanes_below(b, lane);
                g = pi < P ? -gd[(size_t)row * P + pi] * S / (expf(s[j] * invT - M) + S) : 0.f;
            } else {
                g = expf(s[j] * invT - M) * c;
            }
            dsims[(size_t)row * cols + j] = g * invT;
        }
        base += __popcll(b);
    }
}

int fill_params(const vince_infonce_desc* d, InfoParams& p) {
    VINCE_CHECK_ARG(d, VINCE_E_ARG, "vince_infonce: null descriptor");
    VINCE_CHECK_ARG(d->B > 0 && d->Bk >= 0 && d->K >= 0 && (d->Bk + d->K) > 0, VINCE_E_SHAPE, "vince_infonce: bad sizes");
    VINCE_CHECK_ARG(d->D == 64 || d->D == 128, VINCE_E_UNSUPPORTED, "vince_infonce: D=%d unsupported (64 or 128)", d->D);
    VINCE_CHECK_ARG(d->frames >= 1 && d->B % d->frames == 0, VINCE_E_SHAPE, "vince_infonce: B=%d not a multiple of frames=%d",
                    d->B, d->frames);
    VINCE_CHECK_ARG(d->Bk == d->B, VINCE_E_SHAPE,
                    "vince_infonce: in-batch column count %d must equal B=%d (a short final batch would misalign the "
                    "[B | K] mask blocks, vince_model.py:240)", d->Bk, d->B);
    p.d = *d;
    const int rowtiles = (d->B + RT - 1) / RT;
    p.nslabs_inb = (d->Bk + SL - 1) / SL;
    p.nslabs_q = (d->K + SL - 1) / SL;
    p.parts_inb = p.nslabs_inb;
    int target = 512 / rowtiles;
    if (target < 1) target = 1;
    p.slabs_per_part_q = p.nslabs_q > 0 ? (p.nslabs_q + target - 1) / target : 1;
    p.parts_q = p.nslabs_q > 0 ? (p.nslabs_q + p.slabs_per_part_q - 1) / p.slabs_per_part_q : 0;
    return VINCE_OK;
}

}  // namespace

extern "C" size_t vince_infonce_workspace_bytes(const vince_infonce_desc* d) {
    InfoParams p;
    if (fill_params(d, p) != VINCE_OK) return 0;
    return (size_t)3 * (p.parts_inb + p.parts_q) * d->B * sizeof(float);
}

extern "C" int vince_infonce_fwd(const vince_infonce_desc* d, const float* q, const float* inb, const float* queue,
                                 float* pos, float* row_max, float* neg_sum, float* dists, float* softmax_weights,
                                 float* scalars, float* logits, void* workspace, void* stream) {
    InfoParams p;
    int rc = fill_params(d, p);
    if (rc != VINCE_OK) return rc;
    VINCE_CHECK_ARG(q && inb && pos && row_max && neg_sum && dists && softmax_weights && scalars && workspace, VINCE_E_ARG,
                    "vince_infonce_fwd: null pointer");
    VINCE_CHECK_ARG(d->K == 0 || queue, VINCE_E_ARG, "vince_infonce_fwd: queue missing");
    p.q = q; p.inb = inb; p.queue = queue; p.pos = pos; p.part = (float*)workspace;
    p.row_max = nullptr; p.neg_sum = nullptr; p.grad_scale = nullptr; p.dq = nullptr; p.wmat = nullptr; p.logits = logits;
    const int P = p.parts_inb + p.parts_q;
    dim3 grid((d->B + RT - 1) / RT, P);
    if (d->D == 64) hipLaunchKernelGGL(infonce_fwd_partial<64>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(infonce_fwd_partial<128>, grid, dim3(256), 0, (hipStream_t)stream, p);
    VINCE_CHECK_LAUNCH();
    hipLaunchKernelGGL(infonce_merge, dim3(1), dim3(MERGE_THREADS), 0, (hipStream_t)stream, p, P, row_max, neg_sum, dists,
                       softmax_weights, scalars);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_infonce_bwd(const vince_infonce_desc* d, const float* q, const float* inb, const float* queue,
                                 const float* pos, const float* row_max, const float* neg_sum, const float* grad_scale,
                                 const float* logits, float* dq, float* wmat, void* stream) {
    InfoParams p;
    int rc = fill_params(d, p);
    if (rc != VINCE_OK) return rc;
    VINCE_CHECK_ARG(q && inb && pos && row_max && neg_sum && grad_scale && dq, VINCE_E_ARG, "vince_infonce_bwd: null pointer");
    VINCE_CHECK_ARG(d->K == 0 || queue, VINCE_E_ARG, "vince_infonce_bwd: queue missing");
    p.q = q; p.inb = inb; p.queue = queue; p.pos = (float*)pos; p.part = nullptr;
    p.row_max = row_max; p.neg_sum = neg_sum; p.grad_scale = grad_scale; p.dq = dq; p.wmat = wmat; p.logits = (float*)logits;
    const int P = p.parts_inb + p.parts_q;
    dim3 grid((d->B + RT - 1) / RT, P);
    if (d->D == 64) hipLaunchKernelGGL(infonce_bwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(infonce_bwd_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_sce_rows_fwd(const float* sims, const uint8_t* mask, int32_t B, int32_t cols, int32_t P,
                                  float inv_temperature, float* dists, float* softmax_weights, float* row_max,
                                  float* neg_sum, void* stream) {
    VINCE_CHECK_ARG(sims && mask && dists && softmax_weights && row_max && neg_sum && B > 0 && cols > 0 && P > 0, VINCE_E_ARG,
                    "vince_sce_rows_fwd: bad arguments");
    hipLaunchKernelGGL(sce_rows_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, sims, mask, B, cols, P,
                       inv_temperature, dists, softmax_weights, row_max, neg_sum);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_sce_rows_bwd(const float* sims, const uint8_t* mask, int32_t B, int32_t cols, int32_t P,
                                  float inv_temperature, const float* row_max, const float* neg_sum,
                                  const float* grad_dists, float* dsims, void* stream) {
    VINCE_CHECK_ARG(sims && mask && row_max && neg_sum && grad_dists && dsims && B > 0 && cols > 0 && P > 0, VINCE_E_ARG,
                    "vince_sce_rows_bwd: bad arguments");
    hipLaunchKernelGGL(sce_rows_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, sims, mask, B, cols, P,
                       inv_temperature, row_max, neg_sum, grad_dists, dsims);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
