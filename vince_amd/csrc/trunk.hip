// Trunk engine: sequences the ResNet-18/50 trunk (conv1 .. layer4, i.e. the first 8 children that
// models/building_blocks/backbone_models.py:39-54 runs for final_layer=-2) plus the global average pool
// (vince_model.py:33) forward and backward on one HIP stream.  Pure host-side planning + launches of the kernels
// in conv_igemm.hip / conv_wgrad.hip / bn_pool.hip / misc.hip; no device code here.
//
// Memory plan (all inside one caller-provided workspace, sized for 288 GB parts: nothing is recomputed):
//   x0                      input as NHWC with channels padded 3 -> 4 (f32) / 8 (bf16)
//   per conv: y             raw conv output (BatchNorm input), kept for backward
//   per BN+ReLU: a / z      activation, kept for backward (ReLU mask + next conv's wgrad operand)
//   stats / sums            fp64 [C][2] per BN for the forward statistics and the backward reductions
//   consts                  per BN: scale, shift, batch mean, invstd (float[C] each)
//   4 gradient scratch buffers of the largest activation size (dZ, dY, dA, dX roles rotate)
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct ConvL {
    int Ci, Cip, Co, k, stride, pad, Hi, Wi, Ho, Wo;
    int param;          // index into params[]
    size_t wk, wt;      // byte offsets in the weight cache (wt == SIZE_MAX: none)
    std::string name;
};
struct BnL {
    int C, gamma, beta, index;   // param indices + bn index
    int R;                       // statistic replicas the producers spread over (few workgroups -> few replicas)
    size_t stats, sums, consts;  // stats/sums: offset in doubles; consts: offset in floats
    std::string name;
};
struct Blk {
    int nconv;
    ConvL c[3];
    BnL b[3];
    bool has_ds;
    ConvL cd;
    BnL bd;
    size_t x_in, y[3], a[2], yd, z, zmask;   // byte offsets in workspace (zmask: 1 byte per 16-B chunk of z)
    size_t gram, colsum;                      // Gram-statistics scratch of conv3's input (byte offsets; NONE: not eligible)
    size_t alg;                               // BatchNorm-backward algebra scratch (coef, wd, nq, nr: see alg_ptrs), beside gram
    size_t algR;                              // ... and its raw weight gradient R (float[Co][K]), inside the off_algR region
};

constexpr size_t NONE = (size_t)-1;
// Gram-statistics residual join: conv3 reductions up to this length; replicas of the column sums
constexpr int GRAM_R = 4;
// 256 (round 5): layer3's bottlenecks take the Gram route too in NO-GRAD forwards (the key encoder, forward + InfoNCE) -- the Gram
// matrix through the weight-gradient launch (15 + 7 us), vince_bn_gram_finalize with one column of G per thread (14 us), the join in
// the implicit-GEMM kernel's epilogue (77 us) against conv3 with statistics (56) + the separate join pass (~75): the same time
// (5.879 vs 5.865 ms per forward, 22.74 vs 22.77 ms per step), 6 fewer BatchNorm passes and 1.2 GB less HBM traffic per step.
// Training forwards keep the separate passes there (backward reads y3; the streaming join kernel stops at K = 128).
// knob `gram_max_k=128`: the round-4 arrangement (cross-check switch).
int gram_max_k() {
    static const int k = (int)vince_knob("gram_max_k", 256);
    return k;
}
constexpr int NDY_MAX = 8;          // most slots the dY ring of the backward pass can be given (VINCE_KNOBS=dy_slots)

}  // namespace

struct vince_trunk {
    vince_trunk_cfg cfg;
    int esize, CH, Cp;
    int sdtype, cf, cb;   // storage dtype of the activations; dtype of forward / gradient convolution launches (differ for VINCE_F32X3)
    std::vector<Blk> blocks;
    ConvL stem;
    BnL stem_bn;
    int sH, sW, pH, pW;   // stem conv output, pool output
    int sWp;              // padded row length of the packed stem input
    int outC, outH, outW;
    int nparams, nbn;
    std::vector<std::string> pnames;
    std::vector<int> pkind, pbn;
    std::vector<std::array<int, 4>> pshape;
    std::vector<std::string> bnnames;
    std::vector<int> bnC;
    size_t off_x0, off_ystem, off_amax, off_p0, off_stats, off_sums, off_consts, off_g[3], off_dy[NDY_MAX];
    size_t n_stats_doubles, n_consts_floats, max_act, ws_bytes, wc_bytes, off_prep_table;
    size_t off_algR = 0, algR_bytes = 0;      // raw weight gradients of the BatchNorm-backward algebra (scratch, zeroed per backward)
    size_t off_gram = 0, gram_bytes = 0;      // Gram matrices + column sums of the eligible blocks (zeroed per forward)
    std::vector<vince_prep_entry> prep_table[2];   // last uploaded batched weight-prep descriptors (training / folded)
    void* prep_table_dev[2] = {nullptr, nullptr};
    size_t off_fold;      // fold constants (scale, bias per BN channel + ones/zeros) inside a weight cache
    // weight-gradient side stream (created on first backward) + per-slot events of the dY ring
    hipStream_t side = nullptr;
    hipEvent_t ev_dy[NDY_MAX] = {}, ev_wg[NDY_MAX] = {}, ev_join = nullptr, ev_alg = nullptr;
    bool wg_pending[NDY_MAX] = {};
    int ndy = 3;                     // slots of the dY ring in use (knob `dy_slots`, 3 .. NDY_MAX)
    // downsample-branch stream of the 4 stage-entry blocks (backward): its own dY buffer and events
    hipStream_t ds_stream = nullptr;
    hipEvent_t ev_ds_start = nullptr, ev_ds_dy = nullptr, ev_ds_wg = nullptr, ev_ds_done = nullptr;
    size_t off_dyd = 0;
    // slab buffers of the reproducible weight gradient (vince_conv_wgrad_det): [0] launches on the caller's stream, [1] on the side stream
    size_t off_wg_scratch[2] = {0, 0}, wg_scratch_bytes = 0;
    // the last grad-enabled forward took the Gram join WITHOUT storing conv3's output for the eligible blocks (alg_block): its
    // backward must run the BatchNorm-backward algebra for exactly those blocks
    bool fwd_alg = false;
    bool is_twin = false;      // a bf16 handle that some fp32-tensor handle shadows into (vince_trunk_set_shadow): its forward never runs
    // host callback of vince_trunk_backward: invoked right after bucket event e has been recorded (vince_trunk_set_bucket_callback)
    void (*bucket_cb)(int32_t, void*) = nullptr;
    void* bucket_cb_user = nullptr;
    // vince_trunk_set_stem_event: the caller's stream leaves backward WITHOUT waiting for the stem's weight gradient (the last launch of
    // the step, alone on the machine); this event is recorded behind it instead
    hipEvent_t stem_event = nullptr;
    // ... and while that launch may still be running it READS the stem input (off_x0), its dY ring slot and the weight-gradient scratch
    // of the workspace: every entry point that rewrites the workspace makes its stream wait for stem_event first (stem_join)
    bool stem_inflight = false;
    // vince_trunk_set_shadow: the bf16 twin whose workspace this (fp32-tensor) handle's grad-enabled forwards also fill
    vince_trunk* shadow = nullptr;
    void* shadow_ws = nullptr;
};

namespace {

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Planner {
    vince_trunk* t;
    size_t ws = 0, wc = 0, nd = 0, nf = 0;
    size_t alloc_act(size_t elems) {
        size_t off = ws;
        size_t bytes = elems * t->esize;
        if (bytes > t->max_act) t->max_act = bytes;
        ws = align_up(ws + bytes);
        return off;
    }
    ConvL conv(const std::string& name, int Ci, int Co, int k, int stride, int pad, int Hi, int Wi, bool want_wt) {
        ConvL c;
        c.name = name;
        c.Ci = Ci;
        c.Cip = Ci < t->CH ? t->Cp : Ci;
        c.Co = Co; c.k = k; c.stride = stride; c.pad = pad; c.Hi = Hi; c.Wi = Wi;
        c.Ho = (Hi + 2 * pad - k) / stride + 1;
        c.Wo = (Wi + 2 * pad - k) / stride + 1;
        c.param = (int)t->pnames.size();
        t->pnames.push_back(name + ".weight");
        t->pkind.push_back(0);
        t->pbn.push_back(-1);
        t->pshape.push_back({Co, Ci, k, k});
        c.wk = wc;
        // (the 3-channel stem keeps room for either formulation: 49 taps x padded channels, or 7 packed row taps x 32)
        const size_t wk_elems = Ci < t->CH ? std::max((size_t)Co * k * k * c.Cip, (size_t)Co * k * 32) : (size_t)Co * k * k * c.Cip;
        wc = align_up(wc + wk_elems * t->esize);
        if (want_wt) {
            c.wt = wc;
            wc = align_up(wc + (size_t)Ci * k * k * Co * t->esize);
        } else {
            c.wt = NONE;
        }
        return c;
    }
    BnL bn(const std::string& name, int C) {
        BnL b;
        b.name = name;
        b.C = C;
        b.index = (int)t->bnnames.size();
        t->bnnames.push_back(name);
        t->bnC.push_back(C);
        b.gamma = (int)t->pnames.size();
        t->pnames.push_back(name + ".weight");
        t->pkind.push_back(1);
        t->pbn.push_back(b.index);
        t->pshape.push_back({C, 0, 0, 0});
        b.beta = (int)t->pnames.size();
        t->pnames.push_back(name + ".bias");
        t->pkind.push_back(2);
        t->pbn.push_back(b.index);
        t->pshape.push_back({C, 0, 0, 0});
        b.R = VINCE_STATS_REPLICAS;
        b.stats = nd; b.sums = nd;   // sums live in a parallel region of the same size
        nd += (size_t)2 * C * VINCE_STATS_REPLICAS;
        b.consts = nf;
        nf += (size_t)4 * C;
        return b;
    }
};

vince_conv_desc fwd_desc(const vince_trunk* t, const ConvL& c) {
    vince_conv_desc d;
    d.N = t->cfg.N; d.Hi = c.Hi; d.Wi = c.Wi; d.Ci = c.Cip;
    d.Ho = c.Ho; d.Wo = c.Wo; d.Co = c.Co;
    d.sh = d.sw = c.stride;
    d.TA = d.TB = c.k;
    d.dh0 = d.dw0 = -c.pad; d.dhs = d.dws = 1;
    d.wt0 = 0; d.wta = c.k; d.wtb = 1; d.WT = c.k * c.k;
    d.OH = c.Ho; d.OW = c.Wo; d.osh = d.osw = 1; d.oh0 = d.ow0 = 0;
    d.Cs = d.Kw = 0;
    return d;
}

// Replicas of a BatchNorm's fp64 accumulators: enough to keep atomic contention per address low (workgroups of the
// producing conv per replica <= ~512), few enough that the consumers folding them in their prologue read little.
int replicas_for(int64_t rows) {
    const int64_t tiles = (rows + 127) / 128;
    return tiles >= 4096 ? 16 : tiles >= 1024 ? 4 : 1;
}

// The 7x7/s2/p3 stem as 7 packed row taps (vince_conv_desc.Cs): the input is stored [N][H][sWp][4] with 3 zero columns on
// the left, so tap kh of output column wo is the 8 consecutive pixels (32 elements, 16-byte aligned) starting at padded
// column 2*wo -- K = 7 x 32 instead of 49 taps x 8 padded channels, one 64-byte (bf16) LDS-DMA row per tap.
constexpr int STEM_CS = 4, STEM_K = 32, STEM_LEFT = 3;
bool stem_packed() {
    static const bool on = (vince_knob("stem_packed", 1) != 0);
    return on;
}
vince_conv_desc fwd_desc(const vince_trunk* t, const ConvL& c);
vince_conv_desc stem_desc(const vince_trunk* t) {
    const ConvL& c = t->stem;
    if (!stem_packed()) return fwd_desc(t, c);
    vince_conv_desc d;
    d.N = t->cfg.N; d.Hi = c.Hi; d.Wi = t->sWp; d.Ci = STEM_K;
    d.Ho = c.Ho; d.Wo = c.Wo; d.Co = c.Co;
    d.sh = d.sw = 2;
    d.TA = 7; d.TB = 1;
    d.dh0 = -3; d.dhs = 1; d.dw0 = 0; d.dws = 0;
    d.wt0 = 0; d.wta = 1; d.wtb = 0; d.WT = 7;
    d.OH = c.Ho; d.OW = c.Wo; d.osh = d.osw = 1; d.oh0 = d.ow0 = 0;
    d.Cs = STEM_CS; d.Kw = 7;
    return d;
}

// dgrad descriptors (one per output-pixel parity class for stride 2): input = dY, "weights" = W^T [Ci][T][Co]
int dgrad_descs(const vince_trunk* t, const ConvL& c, vince_conv_desc out[4]) {
    int n = 0;
    const int s = c.stride, k = c.k, p = c.pad;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            const int r0 = (ph + p) % s, s0 = (pw + p) % s;
            const int TA = r0 < k ? (k - r0 + s - 1) / s : 0, TB = s0 < k ? (k - s0 + s - 1) / s : 0;
            const int gh = c.Hi > ph ? (c.Hi - ph + s - 1) / s : 0, gw = c.Wi > pw ? (c.Wi - pw + s - 1) / s : 0;
            if (TA == 0 || TB == 0 || gh == 0 || gw == 0) continue;
            vince_conv_desc d;
            d.N = t->cfg.N; d.Hi = c.Ho; d.Wi = c.Wo; d.Ci = c.Co;
            d.Ho = gh; d.Wo = gw; d.Co = c.Ci;
            d.sh = d.sw = 1;
            d.TA = TA; d.TB = TB;
            d.dh0 = (ph + p - r0) / s; d.dhs = -1;
            d.dw0 = (pw + p - s0) / s; d.dws = -1;
            d.wt0 = r0 * k + s0; d.wta = s * k; d.wtb = s; d.WT = k * k;
            d.OH = c.Hi; d.OW = c.Wi; d.osh = d.osw = s; d.oh0 = ph; d.ow0 = pw;
            d.Cs = d.Kw = 0;
            out[n++] = d;
        }
    return n;
}

inline unsigned char* at(void* ws, size_t off) { return (unsigned char*)ws + off; }

}  // namespace

extern "C" int vince_trunk_create(const vince_trunk_cfg* cfg, vince_trunk_t* out) {
    VINCE_CHECK_ARG(cfg && out, VINCE_E_ARG, "vince_trunk_create: null pointer");
    VINCE_CHECK_ARG(cfg->arch == 18 || cfg->arch == 50, VINCE_E_UNSUPPORTED, "vince_trunk_create: arch %d (18 or 50)", cfg->arch);
    VINCE_CHECK_ARG(cfg->dtype == VINCE_F32 || cfg->dtype == VINCE_BF16 || cfg->dtype == VINCE_F32X3 || cfg->dtype == VINCE_F32X3F, VINCE_E_DTYPE,
                    "vince_trunk_create: bad dtype");
    VINCE_CHECK_ARG(cfg->N > 0 && cfg->H >= 8 && cfg->W >= 8, VINCE_E_SHAPE, "vince_trunk_create: bad input shape");
    vince_trunk* t = new vince_trunk();
    t->cfg = *cfg;
    // VINCE_F32X3: fp32 tensors everywhere; only the convolution launches differ (split-half products: IEEE half halves forward,
    // bfloat16 halves for the gradients)
    t->sdtype = cfg->dtype == VINCE_BF16 ? VINCE_BF16 : VINCE_F32;
    // VINCE_F32X3F: the same forward; gradient launches as single bfloat16 products (VINCE_F32X1B), Gram matrices excepted (wgrad_launch)
    const bool x3any = cfg->dtype == VINCE_F32X3 || cfg->dtype == VINCE_F32X3F;
    t->cf = x3any ? VINCE_F32X3H : t->sdtype;
    t->cb = cfg->dtype == VINCE_F32X3 ? VINCE_F32X3B : cfg->dtype == VINCE_F32X3F ? VINCE_F32X1B : t->sdtype;
    t->esize = t->sdtype == VINCE_F32 ? 4 : 2;
    t->CH = t->sdtype == VINCE_F32 ? 4 : 8;
    t->Cp = t->CH;
    t->max_act = 0;
    Planner P{t};
    const int N = cfg->N;
    t->stem = P.conv("conv1", 3, 64, 7, 2, 3, cfg->H, cfg->W, false);
    t->sWp = (2 * t->stem.Wo + 6 + 1) & ~1;   // last tap row ends at padded column 2*(Wo-1) + 8; even for 16-byte rows
    // (room for either stem formulation: VINCE_STEM_PACKED=0 selects the 49-tap one, a measurement / cross-check aid)
    t->off_x0 = P.alloc_act(std::max((size_t)N * cfg->H * t->sWp * STEM_CS, (size_t)N * cfg->H * cfg->W * t->Cp));
    t->stem_bn = P.bn("bn1", 64);   // (R stays at the maximum: 25 088 workgroups feed it at 224 x 224)
    t->sH = t->stem.Ho; t->sW = t->stem.Wo;
    t->pH = (t->sH + 2 - 3) / 2 + 1; t->pW = (t->sW + 2 - 3) / 2 + 1;
    t->off_ystem = P.alloc_act((size_t)N * t->sH * t->sW * 64);
    t->off_p0 = P.alloc_act((size_t)N * t->pH * t->pW * 64);
    t->off_amax = P.ws;
    P.ws = align_up(P.ws + (size_t)N * t->pH * t->pW * 64);

    const bool bottleneck = cfg->arch == 50;
    const int layers18[4] = {2, 2, 2, 2}, layers50[4] = {3, 4, 6, 3};
    const int* layers = bottleneck ? layers50 : layers18;
    const int expansion = bottleneck ? 4 : 1;
    int inpl = 64, H = t->pH, W = t->pW;
    size_t cur = t->off_p0;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        for (int bi = 0; bi < layers[li]; ++bi) {
            const int stride = (li > 0 && bi == 0) ? 2 : 1;
            const std::string pre = "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            const int outp = planes * expansion;
            Blk b;
            b.x_in = cur;
            b.a[0] = b.a[1] = b.yd = NONE;
            b.y[0] = b.y[1] = b.y[2] = NONE;
            if (!bottleneck) {   // resnet.py:53-92
                b.nconv = 2;
                b.c[0] = P.conv(pre + "conv1", inpl, planes, 3, stride, 1, H, W, true);
                b.b[0] = P.bn(pre + "bn1", planes);
                b.c[1] = P.conv(pre + "conv2", planes, planes, 3, 1, 1, b.c[0].Ho, b.c[0].Wo, true);
                b.b[1] = P.bn(pre + "bn2", planes);
            } else {             // resnet.py:95-137, stride on the 3x3
                b.nconv = 3;
                b.c[0] = P.conv(pre + "conv1", inpl, planes, 1, 1, 0, H, W, true);
                b.b[0] = P.bn(pre + "bn1", planes);
                b.c[1] = P.conv(pre + "conv2", planes, planes, 3, stride, 1, H, W, true);
                b.b[1] = P.bn(pre + "bn2", planes);
                b.c[2] = P.conv(pre + "conv3", planes, outp, 1, 1, 0, b.c[1].Ho, b.c[1].Wo, true);
                b.b[2] = P.bn(pre + "bn3", outp);
            }
            b.has_ds = (bi == 0) && (stride != 1 || inpl != outp);   // resnet.py:205-208
            if (b.has_ds) {
                b.cd = P.conv(pre + "downsample.0", inpl, outp, 1, stride, 0, H, W, true);
                b.bd = P.bn(pre + "downsample.1", outp);
            }
            const int Ho = b.c[b.nconv - 1].Ho, Wo = b.c[b.nconv - 1].Wo;
            for (int ci = 0; ci < b.nconv; ++ci) b.b[ci].R = replicas_for((int64_t)N * b.c[ci].Ho * b.c[ci].Wo);
            if (b.has_ds) b.bd.R = replicas_for((int64_t)N * Ho * Wo);
            for (int ci = 0; ci < b.nconv; ++ci) {
                b.y[ci] = P.alloc_act((size_t)N * b.c[ci].Ho * b.c[ci].Wo * b.c[ci].Co);
                if (ci < b.nconv - 1) b.a[ci] = P.alloc_act((size_t)N * b.c[ci].Ho * b.c[ci].Wo * b.c[ci].Co);
            }
            if (b.has_ds) b.yd = P.alloc_act((size_t)N * Ho * Wo * outp);
            b.z = P.alloc_act((size_t)N * Ho * Wo * outp);
            b.zmask = P.ws;
            P.ws = align_up(P.ws + (size_t)N * Ho * Wo * outp / t->CH);
            cur = b.z;
            inpl = outp; H = Ho; W = Wo;
            t->blocks.push_back(b);
        }
    }
    t->outC = inpl; t->outH = H; t->outW = W;
    t->nparams = (int)t->pnames.size();
    t->nbn = (int)t->bnnames.size();
    t->n_stats_doubles = P.nd;
    t->n_consts_floats = P.nf;
    t->off_stats = P.ws; P.ws = align_up(P.ws + P.nd * sizeof(double));
    t->off_sums = P.ws; P.ws = align_up(P.ws + P.nd * sizeof(double));
    t->off_consts = P.ws; P.ws = align_up(P.ws + P.nf * sizeof(float));
    // Gram-statistics scratch (no-grad train-mode forwards, see gram_join_fwd): per eligible bottleneck block a float[w][w]
    // Gram matrix of conv3's input and double[GRAM_R][w] column sums
    t->off_gram = P.ws;
    for (Blk& b : t->blocks) {
        b.gram = b.colsum = NONE;
        if (b.nconv == 3 && b.c[2].Ci <= gram_max_k()) {
            b.gram = P.ws; P.ws = align_up(P.ws + (size_t)b.c[2].Ci * b.c[2].Ci * sizeof(float));
            b.colsum = P.ws; P.ws = align_up(P.ws + (size_t)GRAM_R * b.c[2].Ci * sizeof(double));
        }
    }
    t->gram_bytes = P.ws - t->off_gram;
    for (Blk& b : t->blocks) {   // (not part of the per-forward zeroing: written whole by vince_bn3_bwd_prepare)
        b.alg = NONE;
        if (b.gram != NONE) {
            const size_t w = b.c[2].Ci, co = b.c[2].Co;
            b.alg = P.ws;
            P.ws = align_up(P.ws + align_up(5 * co * sizeof(float)) + align_up(w * 3 * co * 2) + align_up(w * sizeof(float)));
        }
    }
    // the raw weight gradients R = g^T a of the algebra blocks (float[Co][K] each, one contiguous region zeroed once per backward):
    // scratch, so that the gradient buffer itself is only ever ADDED to (gradient accumulation without zero_grad stays correct)
    t->off_algR = P.ws;
    for (Blk& b : t->blocks) {
        b.algR = NONE;
        if (b.alg != NONE) { b.algR = P.ws; P.ws = align_up(P.ws + (size_t)b.c[2].Co * b.c[2].Ci * sizeof(float)); }
    }
    t->algR_bytes = P.ws - t->off_algR;
    for (int i = 0; i < 3; ++i) { t->off_g[i] = P.ws; P.ws = align_up(P.ws + t->max_act); }
    t->ndy = (int)vince_knob("dy_slots", 3);
    if (t->ndy < 3) t->ndy = 3;
    if (t->ndy > NDY_MAX) t->ndy = NDY_MAX;
    for (int i = 0; i < t->ndy; ++i) { t->off_dy[i] = P.ws; P.ws = align_up(P.ws + t->max_act); }
    t->off_dyd = P.ws; P.ws = align_up(P.ws + t->max_act);
    {   // the largest slab set any weight gradient (or Gram matrix) of this trunk wants
        size_t need = 0;
        auto want = [&](const vince_conv_desc& d, int ci_dw) { need = std::max(need, vince_conv_wgrad_scratch_bytes(&d, t->cb, ci_dw)); };
        want(stem_desc(t), 3);
        for (const Blk& b : t->blocks) {
            for (int ci = 0; ci < b.nconv; ++ci) want(fwd_desc(t, b.c[ci]), b.c[ci].Ci);
            if (b.has_ds) want(fwd_desc(t, b.cd), b.cd.Ci);
            if (b.gram != NONE) {
                vince_conv_desc dg = fwd_desc(t, b.c[2]);
                dg.Co = b.c[2].Ci;
                want(dg, b.c[2].Ci);
                need = std::max(need, vince_bn_train_apply_gram_scratch_bytes((int64_t)t->cfg.N * b.c[2].Hi * b.c[2].Wi, b.c[2].Ci));
            }
        }
        t->wg_scratch_bytes = align_up(need);
        for (int i = 0; i < 2; ++i) { t->off_wg_scratch[i] = P.ws; P.ws = align_up(P.ws + t->wg_scratch_bytes); }
    }
    t->ws_bytes = P.ws;
    t->off_prep_table = P.wc;
    t->off_fold = align_up(P.wc + 128 * sizeof(vince_prep_entry));
    t->wc_bytes = align_up(t->off_fold + (2 * (P.nf / 4) + 128) * sizeof(float));
    *out = t;
    return VINCE_OK;
}

// The engine's own streams are PROCESS-WIDE, one pair per device, created on first use and never destroyed: this runtime deals streams
// to its (four) hardware queues in creation order, so an engine instance built later in the process -- a second solver, a trunk for
// another input size -- must land on the queues the first one had, not on whatever the creation counter has reached (measured: the
// same x3 step 52.3 ms in a fresh process, 59.6-60.7 ms as the second solver of a process with per-instance streams).
namespace {
hipStream_t g_side_stream[64] = {}, g_ds_stream[64] = {};
int shared_stream(hipStream_t* pool, bool low_priority, int prio_sign, hipStream_t* out) {
    int dev = 0;
    VINCE_CHECK_HIP(hipGetDevice(&dev));
    VINCE_CHECK_ARG(dev >= 0 && dev < 64, VINCE_E_UNSUPPORTED, "vince_trunk: device index %d", dev);
    if (!pool[dev]) {
        if (low_priority) {
            int least = 0, greatest = 0;
            VINCE_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            VINCE_CHECK_HIP(hipStreamCreateWithPriority(&pool[dev], hipStreamNonBlocking, prio_sign > 0 ? least : greatest));
        } else {
            VINCE_CHECK_HIP(hipStreamCreateWithFlags(&pool[dev], hipStreamNonBlocking));
        }
    }
    *out = pool[dev];
    return VINCE_OK;
}
}  // namespace

extern "C" void vince_trunk_destroy(vince_trunk_t t) {
    if (!t) return;
    if (t->side) {
        hipStreamSynchronize(t->side);     // (the stream itself is shared by every engine instance of the process: it stays)
        for (int i = 0; i < t->ndy; ++i) { hipEventDestroy(t->ev_dy[i]); hipEventDestroy(t->ev_wg[i]); }
        hipEventDestroy(t->ev_join);
        hipEventDestroy(t->ev_alg);
    }
    if (t->ds_stream) {
        hipStreamSynchronize(t->ds_stream);
        hipEventDestroy(t->ev_ds_start); hipEventDestroy(t->ev_ds_dy); hipEventDestroy(t->ev_ds_wg); hipEventDestroy(t->ev_ds_done);
    }
    delete t;
}
extern "C" int vince_trunk_set_bucket_callback(vince_trunk_t t, void (*cb)(int32_t, void*), void* user) {
    VINCE_CHECK_ARG(t, VINCE_E_ARG, "vince_trunk_set_bucket_callback: null handle");
    t->bucket_cb = cb;
    t->bucket_cb_user = user;
    return VINCE_OK;
}
// Deferred stem join, the engine's own protection: the stem's weight gradient of the previous backward may still be reading the
// workspace (the stem input, a dY slot, the scratch).  Whoever rewrites the workspace next waits for it on ITS stream -- a no-op when the
// caller has already joined (FlatSGD / VinceQueueModel wait for the same event), so a C-ABI caller that re-runs forward or backward right
// after a deferred backward cannot race with it.
static int stem_join(vince_trunk* t, void* stream) {
    if (t->stem_inflight && t->stem_event) VINCE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, t->stem_event, 0));
    t->stem_inflight = false;
    return VINCE_OK;
}

extern "C" int vince_trunk_set_shadow(vince_trunk_t t, vince_trunk_t sh, void* shadow_workspace) {
    VINCE_CHECK_ARG(t, VINCE_E_ARG, "vince_trunk_set_shadow: null handle");
    if (!sh) { t->shadow = nullptr; t->shadow_ws = nullptr; return VINCE_OK; }
    VINCE_CHECK_ARG(shadow_workspace && ((uintptr_t)shadow_workspace & 255) == 0, VINCE_E_ALIGN, "vince_trunk_set_shadow: workspace must be 256-byte aligned");
    VINCE_CHECK_ARG(t->sdtype == VINCE_F32 && sh->sdtype == VINCE_BF16 && sh->cf == VINCE_BF16, VINCE_E_DTYPE,
                    "vince_trunk_set_shadow: an fp32-tensor handle shadows into a VINCE_BF16 twin");
    VINCE_CHECK_ARG(sh->cfg.arch == t->cfg.arch && sh->cfg.N == t->cfg.N && sh->cfg.H == t->cfg.H && sh->cfg.W == t->cfg.W &&
                    sh->blocks.size() == t->blocks.size() && sh->n_consts_floats == t->n_consts_floats, VINCE_E_SHAPE,
                    "vince_trunk_set_shadow: the twin must be created for the same architecture and input shape");
    t->shadow = sh;
    t->shadow_ws = shadow_workspace;
    sh->is_twin = true;
    return VINCE_OK;
}
extern "C" int vince_trunk_stem_join(vince_trunk_t t, void* stream) {
    VINCE_CHECK_ARG(t, VINCE_E_ARG, "vince_trunk_stem_join: null handle");
    return stem_join(t, stream);
}
extern "C" int vince_trunk_set_stem_event(vince_trunk_t t, void* event) {
    VINCE_CHECK_ARG(t, VINCE_E_ARG, "vince_trunk_set_stem_event: null handle");
    if (t->stem_inflight && t->stem_event && (hipEvent_t)event != t->stem_event)
        VINCE_CHECK_HIP(hipEventSynchronize(t->stem_event));     // (the event is being replaced or cleared under a running launch: drain it)
    if ((hipEvent_t)event != t->stem_event) t->stem_inflight = false;
    t->stem_event = (hipEvent_t)event;
    return VINCE_OK;
}
extern "C" int32_t vince_trunk_num_params(vince_trunk_t t) { return t ? t->nparams : 0; }
extern "C" int32_t vince_trunk_num_bn(vince_trunk_t t) { return t ? t->nbn : 0; }
extern "C" int32_t vince_trunk_num_blocks(vince_trunk_t t) { return t ? (int32_t)t->blocks.size() : 0; }
extern "C" int32_t vince_trunk_out_channels(vince_trunk_t t) { return t ? t->outC : 0; }
extern "C" int32_t vince_trunk_out_hw(vince_trunk_t t, int32_t* h, int32_t* w) {
    if (!t) return VINCE_E_ARG;
    if (h) *h = t->outH;
    if (w) *w = t->outW;
    return VINCE_OK;
}
extern "C" size_t vince_trunk_workspace_bytes(vince_trunk_t t) { return t ? t->ws_bytes : 0; }
extern "C" size_t vince_trunk_weight_cache_bytes(vince_trunk_t t) { return t ? t->wc_bytes : 0; }

extern "C" int vince_trunk_param_info(vince_trunk_t t, int32_t idx, char* name, int32_t name_cap, int32_t* kind,
                                      int32_t shape[4], int32_t* bn_index) {
    VINCE_CHECK_ARG(t && idx >= 0 && idx < t->nparams, VINCE_E_ARG, "vince_trunk_param_info: bad index %d", idx);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", t->pnames[idx].c_str());
    if (kind) *kind = t->pkind[idx];
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = t->pshape[idx][i];
    if (bn_index) *bn_index = t->pbn[idx];
    return VINCE_OK;
}

extern "C" int vince_trunk_bn_info(vince_trunk_t t, int32_t bn_index, char* name, int32_t name_cap, int32_t* channels) {
    VINCE_CHECK_ARG(t && bn_index >= 0 && bn_index < t->nbn, VINCE_E_ARG, "vince_trunk_bn_info: bad index %d", bn_index);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", t->bnnames[bn_index].c_str());
    if (channels) *channels = t->bnC[bn_index];
    return VINCE_OK;
}

extern "C" void* vince_trunk_input_ptr(vince_trunk_t t, void* workspace, int32_t* row_width, int32_t* left) {
    if (!t || !workspace) return nullptr;
    if (row_width) *row_width = t->sWp;
    if (left) *left = STEM_LEFT;
    return (unsigned char*)workspace + t->off_x0;
}

extern "C" const void* vince_trunk_spatial_ptr(vince_trunk_t t, const void* workspace) {
    if (!t || !workspace) return nullptr;
    return (const unsigned char*)workspace + t->blocks.back().z;
}

namespace {

struct Ctx {
    vince_trunk* t;
    const float* const* params;
    const void* wcache;
    void* ws;
    void* stream;
    int dtype;
    float* consts(const BnL& b, int which) { return (float*)at(ws, t->off_consts) + b.consts + (size_t)which * b.C; }
    double* stats(const BnL& b) { return (double*)at(ws, t->off_stats) + b.stats; }
    double* sums(const BnL& b) { return (double*)at(ws, t->off_sums) + b.sums; }
};

#define RC(expr) do { int _rc = (expr); if (_rc != VINCE_OK) return _rc; } while (0)

// scale / shift / mean / invstd of a BatchNorm -> what a backward over the CENTRED bf16 shadow of its input needs: scale, beta, 0, invstd
__global__ void centre_consts_kernel(const float* __restrict__ src, float* __restrict__ dst, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = src[c], sh = src[C + c], mu = src[2 * C + c];
    dst[c] = sc;
    dst[C + c] = sh + mu * sc;
    dst[2 * C + c] = 0.f;
    dst[3 * C + c] = src[3 * C + c];
}

// The twin's constants of one BatchNorm (float[4][C]: scale, shift, mean, invstd in ITS workspace), and a verbatim copy of this handle's
// (the BatchNorms whose bf16 shadow is the RAW convolution output: the stem's and the downsample branches').
float* twin_consts(vince_trunk* S, void* sw, const BnL& sbn) { return (float*)at(sw, S->off_consts) + sbn.consts; }
int copy_consts(Ctx& c, const BnL& bn, vince_trunk* S, void* sw, const BnL& sbn) {
    VINCE_CHECK_HIP(hipMemcpyAsync(twin_consts(S, sw, sbn), c.consts(bn, 0), (size_t)4 * bn.C * sizeof(float), hipMemcpyDeviceToDevice,
                                   (hipStream_t)c.stream));
    return VINCE_OK;
}

// conv (+ BatchNorm statistics in train mode).  finalize_now: also run the stand-alone finalize -- needed in eval mode and
// where the consumer of scale / shift is not vince_bn_train_apply (the stem's pool, the downsample branch's identity
// affine); everywhere else the finalize rides in the prologue of the apply pass (bn_apply_fwd below).
int conv_bn_fwd(Ctx& c, const ConvL& cv, const BnL& bn, size_t in_off, size_t y_off, float* const* bn_running,
                int64_t* const* bn_nbt, int train_bn, bool finalize_now, const vince_conv_desc* desc = nullptr, void* y2 = nullptr,
                bool in_half_pairs = false) {
    vince_conv_desc d = desc ? *desc : fwd_desc(c.t, cv);
    vince_conv_epi e;
    memset(&e, 0, sizeof(e));
    e.stats = train_bn ? c.stats(bn) : nullptr;
    e.replicas = bn.R;
    if (in_half_pairs) e.flags |= VINCE_EPI_IN_HALF_PAIRS;     // (its input was written as stored half pairs: bn_apply_fwd(..., half_pairs))
    e.out2 = y2;       // (fp32-tensor handles only: the bf16 twin's copy of this raw output, vince_trunk_set_shadow)
    // layer1's expand convolutions (64 -> 256, stride 1: conv3 of every block and the downsample conv) are pure HBM streams that
    // write 4x what they read: the persistent streaming kernel runs them at 4.2 TB/s (122 us) against the implicit-GEMM
    // kernel's 3.1 (166 us), statistics in registers for the whole launch.  At K = 128 (layer2) it does not win (half-line
    // stores, two channel groups re-reading the input): VINCE_XSTATS_MAX_K=128 to try, 0 = off.
    static const int xstats_max_k = VINCE_MEASURE_KNOB("xstats_max_k", 64);
    static const bool strip3x3 = (vince_knob("strip3x3", 1) != 0);
    // (the streaming kernels address their input through one 31-bit buffer descriptor: larger tensors stay on vince_conv_igemm,
    // whose register-staged kernels have no such limit)
    const bool small_in = (unsigned long long)c.t->cfg.N * cv.Hi * cv.Wi * cv.Ci * 2 < 0x7ff00000ull;
    if (!desc && train_bn && c.dtype == VINCE_BF16 && cv.k == 1 && cv.stride == 1 && cv.Ci <= xstats_max_k &&
        (cv.Ci == 64 || cv.Ci == 128) && cv.Co % 256 == 0 && small_in) {
        RC(vince_conv_expand_stats(c.dtype, at(c.ws, in_off), at((void*)c.wcache, cv.wk), (int64_t)c.t->cfg.N * cv.Ho * cv.Wo, cv.Ci,
                                   cv.Co, at(c.ws, y_off), e.stats, e.replicas, c.stream));
    } else if (strip3x3 && !desc && c.dtype == VINCE_BF16 && cv.k == 3 && cv.stride == 1 && cv.Ci == 64 && cv.Co == 64 &&
               cv.Wo == 56 && cv.Ho % 4 == 0 && cv.Hi == cv.Ho && cv.Wi == cv.Wo && small_in) {
        // layer1's 3x3 (conv2 of the 64-wide bottlenecks at 56 x 56): the image-strip kernel (csrc/conv3x3_strip.hip) -- input rows
        // resident in an LDS ring, each element fetched once in whole lines; bit-identical output and statistics
        RC(vince_conv3x3_strip(c.dtype, at(c.ws, in_off), at((void*)c.wcache, cv.wk), c.t->cfg.N, cv.Ho, cv.Wo, cv.Ci, cv.Co, nullptr,
                               at(c.ws, y_off), e.stats, e.replicas, c.stream));
    } else {
        RC(vince_conv_igemm(&d, c.t->cf, at(c.ws, in_off), at((void*)c.wcache, cv.wk), at(c.ws, y_off), &e, c.stream));
    }
    if (finalize_now || !train_bn) {
        const int64_t count = (int64_t)c.t->cfg.N * cv.Ho * cv.Wo;
        RC(vince_bn_finalize(c.stats(bn), count, bn.C, c.params[bn.gamma], c.params[bn.beta], bn_running[2 * bn.index],
                             bn_running[2 * bn.index + 1], bn_nbt ? bn_nbt[bn.index] : nullptr, 0.1f, 1e-5f, train_bn,
                             c.consts(bn, 0), c.consts(bn, 1), c.consts(bn, 2), c.consts(bn, 3), c.stream));
    }
    return VINCE_OK;
}

// out = relu(bn(y) [+ identity affine]); in train mode the BatchNorm's finalize is fused into the same launch
int bn_apply_fwd(Ctx& c, const ConvL& cv, const BnL& bn, size_t y_off, const void* idn, const float* ids, const float* idt,
                 void* out, uint8_t* mask_out, float* const* bn_running, int64_t* const* bn_nbt, int train_bn,
                 double* out_sum = nullptr, void* out2 = nullptr, uint8_t* mask2 = nullptr, void* y2c = nullptr, float* consts2 = nullptr,
                 bool half_pairs = false) {
    const int64_t rows = (int64_t)c.t->cfg.N * cv.Ho * cv.Wo;
    static const bool fuse_fin = (VINCE_MEASURE_KNOB("fuse_finalize", 1) != 0);
    if (!train_bn)
        return vince_bn_apply(c.dtype, at(c.ws, y_off), c.consts(bn, 0), c.consts(bn, 1), idn, ids, idt, out, mask_out, rows,
                              cv.Co, 1, c.stream);
    if (!fuse_fin && !out_sum) {   // measurement aid: the two launches of the unfused path
        RC(vince_bn_finalize(c.stats(bn), rows, bn.C, c.params[bn.gamma], c.params[bn.beta], bn_running[2 * bn.index],
                             bn_running[2 * bn.index + 1], bn_nbt ? bn_nbt[bn.index] : nullptr, 0.1f, 1e-5f, 1,
                             c.consts(bn, 0), c.consts(bn, 1), c.consts(bn, 2), c.consts(bn, 3), c.stream));
        return vince_bn_apply(c.dtype, at(c.ws, y_off), c.consts(bn, 0), c.consts(bn, 1), idn, ids, idt, out, mask_out, rows,
                              cv.Co, 1, c.stream);
    }
    vince_bn_train bt;
    memset(&bt, 0, sizeof(bt));
    bt.stats = c.stats(bn);
    bt.replicas = bn.R;
    bt.count = rows;
    bt.gamma = c.params[bn.gamma];
    bt.beta = c.params[bn.beta];
    bt.running_mean = bn_running[2 * bn.index];
    bt.running_var = bn_running[2 * bn.index + 1];
    bt.num_batches_tracked = bn_nbt ? bn_nbt[bn.index] : nullptr;
    bt.momentum = 0.1f;
    bt.eps = 1e-5f;
    bt.scale = c.consts(bn, 0);
    bt.shift = c.consts(bn, 1);
    bt.save_mean = c.consts(bn, 2);
    bt.save_invstd = c.consts(bn, 3);
    bt.out_sum = out_sum;
    bt.out_sum_replicas = GRAM_R;
    bt.out_bf16 = out2;
    bt.mask_bf16 = mask2;
    bt.y_centred_bf16 = y2c;
    bt.shadow_consts = consts2;
    bt.out_half_pairs = half_pairs ? 1 : 0;
    return vince_bn_train_apply(c.dtype, at(c.ws, y_off), &bt, idn, ids, idt, out, mask_out, rows, cv.Co, 1, c.stream);
}

// the same for the pass that writes conv3's input in a Gram block: train-mode BatchNorm + ReLU + column sums + the Gram matrix of what
// it stores, in one launch (csrc/bn_gram.hip)
int bn_apply_gram_fwd(Ctx& c, const ConvL& cv, const BnL& bn, size_t y_off, void* out, float* const* bn_running, int64_t* const* bn_nbt,
                      double* out_sum, float* gram, void* scratch, size_t scratch_bytes) {
    const int64_t rows = (int64_t)c.t->cfg.N * cv.Ho * cv.Wo;
    vince_bn_train bt;
    memset(&bt, 0, sizeof(bt));
    bt.stats = c.stats(bn);
    bt.replicas = bn.R;
    bt.count = rows;
    bt.gamma = c.params[bn.gamma];
    bt.beta = c.params[bn.beta];
    bt.running_mean = bn_running[2 * bn.index];
    bt.running_var = bn_running[2 * bn.index + 1];
    bt.num_batches_tracked = bn_nbt ? bn_nbt[bn.index] : nullptr;
    bt.momentum = 0.1f;
    bt.eps = 1e-5f;
    bt.scale = c.consts(bn, 0);
    bt.shift = c.consts(bn, 1);
    bt.save_mean = c.consts(bn, 2);
    bt.save_invstd = c.consts(bn, 3);
    bt.out_sum = out_sum;
    bt.out_sum_replicas = GRAM_R;
    return vince_bn_train_apply_gram(c.dtype, at(c.ws, y_off), &bt, out, rows, cv.Co, gram, scratch, scratch_bytes, c.stream);
}

// BatchNorm-backward algebra (csrc/bn_algebra.hip): bf16 bottlenecks whose conv3 reduction fits the Gram scratch and the streaming
// join kernel, every block but the last (whose output gradient arrives from the pool, not from a dgrad epilogue).
// VINCE_BN3_ALGEBRA=0 (read per call, so one process can compare both routes): the separate BatchNorm-backward passes everywhere.
bool alg_env() {
    return vince_knob_live("bn3_algebra", 1) != 0;
}
bool alg_block(const vince_trunk* t, size_t bi) {
    const Blk& b = t->blocks[bi];
    return t->sdtype == VINCE_BF16 && b.nconv == 3 && b.gram != NONE && b.alg != NONE && bi + 1 < t->blocks.size() &&
           (b.c[2].Ci == 64 || b.c[2].Ci == 128) && b.c[2].Co % 256 == 0 && b.c[2].k == 1 && b.c[2].stride == 1 &&
           (unsigned long long)t->cfg.N * b.c[2].Hi * b.c[2].Wi * b.c[2].Ci * 2 < 0x7ff00000ull;
}
struct AlgPtrs { float* coef; void* w2; float* nr; };   // w2: bf16 [w][2][4w] -- tap 0 = wd, tap 1 = nq in its first w entries -- or, split into
                                                        // hi + lo parts, [w][3][4w]: wd_hi, wd_lo, [nq_hi | nq_lo] in the first 2w entries of the third
AlgPtrs alg_ptrs(void* ws, const Blk& b) {
    const size_t w = b.c[2].Ci, co = b.c[2].Co;
    unsigned char* base = (unsigned char*)at(ws, b.alg);
    AlgPtrs a;
    a.coef = (float*)base; base += align_up(5 * co * sizeof(float));
    a.w2 = base; base += align_up(w * 3 * co * 2);
    a.nr = (float*)base;
    return a;
}

// out_mask / gsums: the epilogue of the BatchNorm-backward algebra (vince_conv_epi.out_mask): the stored gradient is gated by the
// ReLU bits of the block BELOW and its per-channel sums land in gsums
int dgrad(Ctx& c, const ConvL& cv, const void* dy, void* dx, bool accumulate, const uint8_t* acc_mask = nullptr,
          const vince_bn_reduce* bnred = nullptr, int replicas = 0, const uint8_t* out_mask = nullptr, double* gsums = nullptr) {
    // block-input gradients of layer1 / layer2 bottlenecks (conv1 = 1x1 stride 1, 4w -> w with w = 64 / 128): the expand shape
    // again, through the persistent streaming kernel (299 vs 362 us, 306 vs 383 us, 177 vs 192 us).  VINCE_XDGRAD=0: off.
    static const bool xdgrad_env = (vince_knob("xdgrad", 1) != 0);
    if (xdgrad_env && accumulate && c.dtype == VINCE_BF16 && cv.k == 1 && cv.stride == 1 && (cv.Co == 64 || cv.Co == 128) &&
        cv.Ci % 256 == 0 && !(bnred && bnred->mask_scale) &&
        (unsigned long long)c.t->cfg.N * cv.Hi * cv.Wi * cv.Co * 2 < 0x7ff00000ull) {
        if (out_mask)
            return vince_conv_expand_dgrad_masked(c.dtype, dy, at((void*)c.wcache, cv.wt), (int64_t)c.t->cfg.N * cv.Hi * cv.Wi, cv.Co, cv.Ci,
                                                  dx, 1, acc_mask, out_mask, gsums, replicas, c.stream);
        return vince_conv_expand_dgrad(c.dtype, dy, at((void*)c.wcache, cv.wt), (int64_t)c.t->cfg.N * cv.Hi * cv.Wi, cv.Co, cv.Ci, dx, 1,
                                       acc_mask, bnred, replicas, c.stream);
    }
    // layer1's 3x3 (64 -> 64 at 56 x 56, stride 1): the image-strip kernel with the taps flipped (csrc/conv3x3_strip.hip) -- every dy
    // element crosses L2 -> LDS once instead of nine times, the BatchNorm-backward reduction of the BatchNorm below rides in its
    // epilogue as in the implicit-GEMM gradient launch.  `strip3x3_dgrad=0`: off (cross-check switch).
    static const bool strip_dgrad = (vince_knob("strip3x3", 1) != 0) && (vince_knob("strip3x3_dgrad", 1) != 0);
    if (strip_dgrad && c.dtype == VINCE_BF16 && !accumulate && !acc_mask && !out_mask && !gsums && cv.k == 3 && cv.stride == 1 &&
        cv.Ci == 64 && cv.Co == 64 && cv.Wi == 56 && cv.Hi % 4 == 0 && cv.Hi == cv.Ho && cv.Wi == cv.Wo &&
        (!bnred || (bnred->mask_scale && !bnred->mask_bits)) &&
        (unsigned long long)c.t->cfg.N * cv.Hi * cv.Wi * cv.Co * 2 < 0x7ff00000ull)
        return vince_conv3x3_strip_dgrad(c.dtype, dy, at((void*)c.wcache, cv.wt), c.t->cfg.N, cv.Hi, cv.Wi, cv.Ci, dx, bnred, replicas, c.stream);
    vince_conv_desc ds[4];
    const int n = dgrad_descs(c.t, cv, ds);
    const int classes = cv.stride * cv.stride;
    if (!accumulate && n < classes)   // some pixel classes receive no gradient (1x1 stride 2): zero them
        RC(vince_zero_async(dx, (size_t)c.t->cfg.N * cv.Hi * cv.Wi * cv.Ci * c.t->esize, c.stream));
    vince_conv_epi e;
    memset(&e, 0, sizeof(e));
    e.flags = accumulate ? VINCE_EPI_ACCUMULATE : 0;
    e.acc_mask = acc_mask;
    if (bnred) e.bnred = *bnred;
    e.replicas = replicas;
    e.out_mask = out_mask;
    e.stats = gsums;
    for (int i = 0; i < n; ++i) RC(vince_conv_igemm(&ds[i], c.t->cb, dy, at((void*)c.wcache, cv.wt), dx, &e, c.stream));
    return VINCE_OK;
}

// Weight-gradient launches.  VINCE_KNOBS wgrad_det: 1 (default) = the GRAM matrices (in = dy = a: BatchNorm constants of the fused joins and
// of the BatchNorm-backward algebra) take the reproducible path -- per-split slabs in the workspace + a fixed-order reduction, so the
// forward is bit-reproducible at no measurable cost; 2 = every weight gradient does (+0.45 ms per step: the slabs are extra HBM traffic,
// 25.38 against 24.92 ms); 0 = fp32 atomics everywhere.  which: 0 = a launch on the caller's stream, 1 = on the engine's side stream
// (each has its own slab buffer).
int wgrad_launch(vince_trunk* t, void* ws, int dtype, const vince_conv_desc& d, const void* in, const void* dy, float* dw, int ci_dw,
                 int which, void* stream) {
    static const int det = (int)vince_knob("wgrad_det", 1);
    const bool gram = in == dy;
    if (gram && dtype == VINCE_F32X1B) dtype = VINCE_F32X3B;     // Gram matrices feed FORWARD statistics: split-half products in every x3 mode
    if (!(det >= 2 || (det == 1 && gram)) || !t->wg_scratch_bytes) return vince_conv_wgrad(&d, dtype, in, dy, dw, ci_dw, 0, stream);
    return vince_conv_wgrad_det(&d, dtype, in, dy, dw, ci_dw, at(ws, t->off_wg_scratch[which]), t->wg_scratch_bytes, stream);
}

int wgrad(Ctx& c, const ConvL& cv, const void* in, const void* dy, float* dw) {
    vince_conv_desc d = fwd_desc(c.t, cv);
    return wgrad_launch(c.t, c.ws, c.t->cb, d, in, dy, dw, cv.Ci, 0, c.stream);
}

// BN backward for y (conv output) given the gradient dz wrt the activation that followed this BatchNorm.
// ReLU mask: `bits` (residual outputs: bytes written by the forward bn_apply), `self_mask` (plain BN+ReLU: the sign
// of y*scale+shift is recomputed from y, which is read anyway), or none (no ReLU: the stem's pooled gradient).
// `reduced`: the (sum g, sum g*xhat) pass already ran inside the epilogue of the dgrad that produced dz (bn_reduce_of).
int bn_bwd(Ctx& c, const BnL& bn, const void* dz, const uint8_t* bits, bool self_mask, size_t y_off, int64_t rows, void* dy,
           void* g_out, float* const* grads, bool reduced = false, const vince_bn_reduce2* second = nullptr) {
    const float* msc = self_mask ? c.consts(bn, 0) : nullptr;
    const float* msh = self_mask ? c.consts(bn, 1) : nullptr;
    if (!reduced)
        RC(vince_bn_bwd_reduce(c.dtype, dz, nullptr, bits, msc, msh, at(c.ws, y_off), c.consts(bn, 2), c.consts(bn, 3),
                               c.sums(bn), rows, bn.C, bn.R, c.stream));
    RC(vince_bn_bwd_apply(c.dtype, dz, nullptr, bits, msc, msh, at(c.ws, y_off), c.consts(bn, 2), c.consts(bn, 3),
                          c.params[bn.gamma], c.sums(bn), rows, dy, g_out, grads[bn.gamma], grads[bn.beta], rows, bn.C,
                          bn.R, second, c.stream));
    return VINCE_OK;
}

vince_bn_reduce bn_reduce_of(Ctx& c, const BnL& bn, const uint8_t* bits, bool self_mask, size_t y_off) {
    vince_bn_reduce r;
    r.y = at(c.ws, y_off);
    r.mask_bits = bits;
    r.mask_scale = self_mask ? c.consts(bn, 0) : nullptr;
    r.mask_shift = self_mask ? c.consts(bn, 1) : nullptr;
    r.mean = c.consts(bn, 2);
    r.invstd = c.consts(bn, 3);
    r.sums = c.sums(bn);
    return r;
}

}  // namespace

namespace {

// Folded inference forward: a bottleneck's conv3 + bias + identity + ReLU through the persistent streaming kernel (csrc/conv_xjoin.hip)
// where it applies -- bf16, K = 64 / 128, Co multiple of 256: layer1 / layer2 -- with bn3's scale as the kernel's out_scale instead of
// folded into the weights (round 5: 4.4-5.2 TB/s against the implicit-GEMM join epilogue's 3.4).  `xjoin=0`: the epilogue everywhere.
// `xjoin_next128=0`: the layer1 -> layer2 transition keeps its own conv1 launch (cross-check / A-B switch)
bool next_env128() {
    return vince_knob_live("xjoin_next128", 1) != 0;
}
// Inference cache: which blocks keep conv3's weights UNFOLDED (bn3's scale goes in as the join's out_scale).  A property of the
// ARCHITECTURE and the dtype only -- one inference cache serves every trunk of a model (VinceModel._wcache_folded), whatever its batch --
// so it must not depend on cfg.N; whether the streaming kernel can take the launch (31-bit descriptor offsets) is asked separately and a
// launch past the limit runs the implicit-GEMM join epilogue with the same out_scale.
bool folded_xjoin_block(const vince_trunk* t, const Blk& b) {
    static const bool on = (vince_knob("xjoin", 1) != 0) && (vince_knob("xjoin_folded", 1) != 0);
    return on && t->sdtype == VINCE_BF16 && t->cf == VINCE_BF16 && b.nconv == 3 && (b.c[2].Ci == 64 || b.c[2].Ci == 128) &&
           b.c[2].Co % 256 == 0 && b.c[2].k == 1 && b.c[2].stride == 1;
}
bool folded_xjoin_fits(const vince_trunk* t, const Blk& b) {
    // (`xjoin_folded_max_bytes`: cross-check switch of the tests -- the fallback is otherwise reached only by batches of thousands of frames)
    const unsigned long long limit = (unsigned long long)vince_knob_live("xjoin_folded_max_bytes", 0x7ff00000L);
    return (unsigned long long)t->cfg.N * b.c[2].Hi * b.c[2].Wi * b.c[2].Ci * 2 < limit;
}

// Batched weight prep shared by the training cache (scale == nullptr) and the BatchNorm-folded inference cache
// (fold_scale: float[sum of BN channels] in BN order, inside the cache).  `which` selects the cached descriptor table.
// part: 0 = every layer, 1 = every layer but the stem (table entry 0), 2 = the stem alone (vince_trunk_prepare_weights_part)
int prepare_common(vince_trunk* t, const float* const* params, void* wcache, const float* fold_scale, int which, void* stream, int part = 0) {
    std::vector<vince_prep_entry> tab;
    auto scale_of = [&](const BnL& b) -> const float* { return fold_scale ? fold_scale + b.consts / 4 : nullptr; };
    auto add = [&](const ConvL& c, const BnL& b) {
        vince_prep_entry e;
        e.w = params[c.param];
        e.wk = at(wcache, c.wk);
        e.wt = (c.wt == NONE || fold_scale) ? nullptr : at(wcache, c.wt);   // inference needs no dgrad copy
        e.scale = scale_of(b);
        e.Co = c.Co; e.T = c.k * c.k; e.Ci = c.Ci; e.Cip = c.Cip; e.Cs = e.Kw = 0;
        tab.push_back(e);
    };
    {   // stem: packed row taps
        vince_prep_entry e;
        e.w = params[t->stem.param];
        e.wk = at(wcache, t->stem.wk);
        e.wt = nullptr;
        e.scale = scale_of(t->stem_bn);
        e.Co = t->stem.Co; e.T = 7; e.Ci = 3; e.Cip = STEM_K; e.Cs = STEM_CS; e.Kw = 7;
        if (!stem_packed()) { e.T = 49; e.Cip = t->stem.Cip; e.Cs = e.Kw = 0; }
        tab.push_back(e);
    }
    for (const Blk& b : t->blocks) {
        for (int ci = 0; ci < b.nconv; ++ci) {
            add(b.c[ci], b.b[ci]);
            // (the streaming join multiplies by bn3's scale itself: its conv3 weights stay unfolded in the inference cache)
            if (fold_scale && ci == b.nconv - 1 && folded_xjoin_block(t, b)) tab.back().scale = nullptr;
        }
        if (b.has_ds) add(b.cd, b.bd);
    }
    void* dev_table = at(wcache, t->off_prep_table);
    // the table only changes when the caller's buffers move: upload it then (a pageable H2D copy synchronises the host
    // with the stream, which must not happen every step)
    std::vector<vince_prep_entry>& last = t->prep_table[which];
    const bool same = t->prep_table_dev[which] == dev_table && last.size() == tab.size() &&
                      memcmp(last.data(), tab.data(), tab.size() * sizeof(vince_prep_entry)) == 0;
    if (!same) {
        VINCE_CHECK_HIP(hipMemcpyAsync(dev_table, tab.data(), tab.size() * sizeof(vince_prep_entry), hipMemcpyHostToDevice,
                                       (hipStream_t)stream));
        VINCE_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        last = tab;
        t->prep_table_dev[which] = dev_table;
    }
    // (VINCE_F32X3: the split-half weight layout -- IEEE half pairs in the forward copies, bfloat16 pairs in the transposed ones)
    const vince_prep_entry* first = (const vince_prep_entry*)dev_table + (part == 1 ? 1 : 0);
    const int32_t count = part == 0 ? (int32_t)tab.size() : part == 1 ? (int32_t)tab.size() - 1 : 1;
    if (count <= 0) return VINCE_OK;
    return vince_prepare_weights_batched(t->cf == VINCE_F32X3H ? VINCE_F32X3 : t->sdtype, first, count, stream);
}

// fold constants inside a folded weight cache: scale[c] and bias[c] per BN channel (BN order, offset b.consts / 4), then
// 64 ones and 64 zeros (the stem pool's identity affine)
inline float* fold_scale(vince_trunk* t, void* wcache) { return (float*)at(wcache, t->off_fold); }
inline float* fold_bias(vince_trunk* t, void* wcache) { return fold_scale(t, wcache) + t->n_consts_floats / 4; }
inline float* fold_ones(vince_trunk* t, void* wcache) { return fold_bias(t, wcache) + t->n_consts_floats / 4; }

}  // namespace

extern "C" int vince_trunk_prepare_weights(vince_trunk_t t, const float* const* params, void* wcache, void* stream) {
    VINCE_CHECK_ARG(t && params && wcache, VINCE_E_ARG, "vince_trunk_prepare_weights: null pointer");
    return prepare_common(t, params, wcache, nullptr, 0, stream);
}

extern "C" int vince_trunk_prepare_weights_part(vince_trunk_t t, const float* const* params, void* wcache, int32_t part, void* stream) {
    VINCE_CHECK_ARG(t && params && wcache, VINCE_E_ARG, "vince_trunk_prepare_weights_part: null pointer");
    VINCE_CHECK_ARG(part >= 0 && part <= 2, VINCE_E_ARG, "vince_trunk_prepare_weights_part: part %d (0 all, 1 all but the stem, 2 the stem)", part);
    return prepare_common(t, params, wcache, nullptr, 0, stream, part);
}

extern "C" int vince_trunk_prepare_weights_folded(vince_trunk_t t, const float* const* params, float* const* bn_running,
                                                  void* wcache, void* stream) {
    VINCE_CHECK_ARG(t && params && bn_running && wcache, VINCE_E_ARG, "vince_trunk_prepare_weights_folded: null pointer");
    float* sc = fold_scale(t, wcache);
    float* bi = fold_bias(t, wcache);
    auto fold = [&](const BnL& b) -> int {   // eval-mode finalize: scale = gamma / sqrt(var + eps), bias = beta - mean * scale
        return vince_bn_finalize(nullptr, 1, b.C, params[b.gamma], params[b.beta], bn_running[2 * b.index],
                                 bn_running[2 * b.index + 1], nullptr, 0.1f, 1e-5f, 0, sc + b.consts / 4, bi + b.consts / 4,
                                 nullptr, nullptr, stream);
    };
    RC(fold(t->stem_bn));
    for (const Blk& b : t->blocks) {
        for (int ci = 0; ci < b.nconv; ++ci) RC(fold(b.b[ci]));
        if (b.has_ds) RC(fold(b.bd));
    }
    RC(vince_fill_f32_async(fold_ones(t, wcache), 64, 1.f, stream));
    RC(vince_fill_f32_async(fold_ones(t, wcache) + 64, 64, 0.f, stream));
    return prepare_common(t, params, wcache, sc, 1, stream);
}

extern "C" int vince_trunk_forward_folded(vince_trunk_t t, const void* wcache, const float* input, const int64_t* perm,
                                          int32_t jig_h, int32_t jig_w, void* workspace, float* pooled, void* stream) {
    VINCE_CHECK_ARG(t && wcache && workspace && pooled, VINCE_E_ARG, "vince_trunk_forward_folded: null pointer");
    VINCE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wcache & 255) == 0, VINCE_E_ALIGN,
                    "vince_trunk_forward_folded: workspace / weight cache must be 256-byte aligned");
    const int N = t->cfg.N, dtype = t->sdtype;
    RC(stem_join(t, stream));
    const float* bias = fold_bias(t, (void*)wcache);
    const float* ones = fold_ones(t, (void*)wcache);
    if (!input) {   // staged input, see vince_trunk_forward
        VINCE_CHECK_ARG(stem_packed(), VINCE_E_UNSUPPORTED, "vince_trunk_forward_folded: staged input needs the packed stem layout");
    } else if (!stem_packed()) {
        if (jig_h > 0)
            RC(vince_jigsaw_nchw_to_nhwc(dtype, input, at(workspace, t->off_x0), N / 9, 3, jig_h, jig_w, t->cfg.H, t->cfg.W,
                                         t->Cp, stream));
        else
            RC(vince_input_nchw_to_nhwc(dtype, input, perm, at(workspace, t->off_x0), N, 3, t->cfg.H, t->cfg.W, t->Cp, stream));
    } else if (jig_h > 0) {
        VINCE_CHECK_ARG(N % 9 == 0, VINCE_E_SHAPE, "vince_trunk_forward_folded: jigsaw needs N multiple of 9");
        RC(vince_jigsaw_nchw_to_rows(dtype, input, at(workspace, t->off_x0), N / 9, 3, jig_h, jig_w, t->cfg.H, t->cfg.W,
                                     t->sWp, STEM_LEFT, stream));
    } else {
        RC(vince_input_nchw_to_rows(dtype, input, perm, at(workspace, t->off_x0), N, 3, t->cfg.H, t->cfg.W, t->sWp, STEM_LEFT,
                                    stream));
    }
    // conv + bias [+ ReLU] [+ residual join: out = relu(conv + bias + out)]
    auto conv = [&](const vince_conv_desc& d, const ConvL& cv, const BnL& b, const void* in, void* out, int flags) -> int {
        vince_conv_epi e;
        memset(&e, 0, sizeof(e));
        e.flags = flags;
        e.bias = bias + b.consts / 4;
        return vince_conv_igemm(&d, t->cf, in, at((void*)wcache, cv.wk), out, &e, stream);
    };
    RC(conv(stem_desc(t), t->stem, t->stem_bn, at(workspace, t->off_x0), at(workspace, t->off_ystem), 0));
    RC(vince_stem_pool_fwd(dtype, at(workspace, t->off_ystem), ones, ones + 64, at(workspace, t->off_p0),
                           (uint8_t*)at(workspace, t->off_amax), N, t->sH, t->sW, 64, stream));
    void* cur = at(workspace, t->off_p0);   // the running block output; identity blocks update it in place
    static const bool next_folded = vince_knob("xjoin_next", 1) != 0;
    bool conv1_done = false;   // block bi's conv1 (+ bias + ReLU) came out of block bi-1's join launch (vince_conv_expand_join_next)
    for (size_t bi = 0; bi < t->blocks.size(); ++bi) {
        const Blk& b = t->blocks[bi];
        const void* in = cur;
        const bool skip_conv1 = conv1_done;
        conv1_done = false;
        for (int ci = 0; ci < b.nconv - 1; ++ci) {
            const ConvL& cv = b.c[ci];
            if (ci == 0 && skip_conv1) {
                in = at(workspace, b.a[0]);
                continue;
            }
            // layer1's 3x3 (64 -> 64 at 56 x 56, bf16): the image-strip kernel with the bias + ReLU epilogue (bit-identical output)
            static const bool strip3x3 = (vince_knob("strip3x3", 1) != 0) && (vince_knob("xjoin_folded", 1) != 0);
            if (strip3x3 && dtype == VINCE_BF16 && t->cf == VINCE_BF16 && cv.k == 3 && cv.stride == 1 && cv.Ci == 64 && cv.Co == 64 &&
                cv.Wo == 56 && cv.Ho % 4 == 0 && cv.Hi == cv.Ho && cv.Wi == cv.Wo &&
                (unsigned long long)N * cv.Hi * cv.Wi * cv.Ci * 2 < 0x7ff00000ull) {
                RC(vince_conv3x3_strip_bias(dtype, in, at((void*)wcache, cv.wk), N, cv.Ho, cv.Wo, cv.Ci, cv.Co, bias + b.b[ci].consts / 4, 1,
                                            at(workspace, b.a[ci]), stream));
            } else {
                RC(conv(fwd_desc(t, cv), cv, b.b[ci], in, at(workspace, b.a[ci]), VINCE_EPI_RELU));
            }
            in = at(workspace, b.a[ci]);
        }
        const int L = b.nconv - 1;
        void* out = cur;
        if (b.has_ds) {   // the downsample branch (bias, no ReLU) lands in z, then conv_L joins onto it
            out = at(workspace, b.z);
            RC(conv(fwd_desc(t, b.cd), b.cd, b.bd, cur, out, 0));
        } else if (bi + 1 == t->blocks.size()) {   // the trunk output is read at blocks.back().z
            out = at(workspace, b.z);
            VINCE_CHECK_HIP(hipMemcpyAsync(out, cur, (size_t)N * b.c[L].Ho * b.c[L].Wo * b.c[L].Co * t->esize,
                                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
        }
        const Blk* nb = bi + 1 < t->blocks.size() ? &t->blocks[bi + 1] : nullptr;
        if (folded_xjoin_block(t, b) && folded_xjoin_fits(t, b) && next_folded && b.c[L].Ci == 64 && b.c[L].Co == 256 && nb && nb->nconv == 3 &&
            nb->c[0].k == 1 && nb->c[0].stride == 1 && nb->c[0].Ci == 256 &&
            ((nb->c[0].Co == 64 && !nb->has_ds) || (nb->c[0].Co == 128 && next_env128()))) {
            // ... and the next block's conv1 + bias + ReLU on the block output while it is in LDS
            const ConvL& cv = b.c[L];
            RC(vince_conv_expand_join_next(dtype, in, at((void*)wcache, cv.wk), (int64_t)N * cv.Ho * cv.Wo, cv.Ci, cv.Co,
                                           fold_scale(t, (void*)wcache) + b.b[L].consts / 4, bias + b.b[L].consts / 4, out, nullptr, nullptr, out,
                                           nullptr, nullptr, 1, at((void*)wcache, nb->c[0].wk), nb->c[0].Co, at(workspace, nb->a[0]), nullptr, 0,
                                           bias + nb->b[0].consts / 4, 1, stream));
            conv1_done = true;
        } else if (folded_xjoin_block(t, b) && folded_xjoin_fits(t, b)) {
            const ConvL& cv = b.c[L];
            RC(vince_conv_expand_join(dtype, in, at((void*)wcache, cv.wk), (int64_t)N * cv.Ho * cv.Wo, cv.Ci, cv.Co,
                                      fold_scale(t, (void*)wcache) + b.b[L].consts / 4, bias + b.b[L].consts / 4, out, nullptr, nullptr, out,
                                      nullptr, nullptr, 1, stream));
        } else if (folded_xjoin_block(t, b)) {
            // the cache holds conv3 UNFOLDED for this block, the tensor is past the streaming kernel's 31-bit offsets: the implicit-GEMM
            // join epilogue with bn3's scale as out_scale (out = relu(conv * scale + bias + out))
            vince_conv_epi e;
            memset(&e, 0, sizeof(e));
            e.flags = VINCE_EPI_ACCUMULATE | VINCE_EPI_RELU;
            e.out_scale = fold_scale(t, (void*)wcache) + b.b[L].consts / 4;
            e.bias = bias + b.b[L].consts / 4;
            const vince_conv_desc d3 = fwd_desc(t, b.c[L]);
            RC(vince_conv_igemm(&d3, t->cf, in, at((void*)wcache, b.c[L].wk), out, &e, stream));
        } else {
            RC(conv(fwd_desc(t, b.c[L]), b.c[L], b.b[L], in, out, VINCE_EPI_ACCUMULATE | VINCE_EPI_RELU));
        }
        cur = out;
    }
    RC(vince_avgpool_fwd(dtype, cur, pooled, N, t->outH * t->outW, t->outC, stream));
    return VINCE_OK;
}

extern "C" int vince_trunk_forward(vince_trunk_t t, const float* const* params, const void* wcache, float* const* bn_running,
                                   int64_t* const* bn_nbt, const float* input, const int64_t* perm, int32_t jig_h,
                                   int32_t jig_w, void* workspace, float* pooled, int32_t train_bn, int32_t save,
                                   void* stream) {
    VINCE_CHECK_ARG(t && params && wcache && bn_running && workspace && pooled, VINCE_E_ARG,
                    "vince_trunk_forward: null pointer");
    VINCE_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wcache & 255) == 0, VINCE_E_ALIGN,
                    "vince_trunk_forward: workspace / weight cache must be 256-byte aligned");
    Ctx c{t, params, wcache, workspace, stream, t->sdtype};
    const int N = t->cfg.N;
    RC(stem_join(t, stream));
    if (train_bn)
        RC(vince_zero_async(at(workspace, t->off_stats), t->n_stats_doubles * sizeof(double), stream));
    if (!input) {
        // staged input: the caller has already written the stem layout at vince_trunk_input_ptr() (e.g. straight from
        // uint8 frames with vince_input_u8hwc_to_rows)
        VINCE_CHECK_ARG(stem_packed(), VINCE_E_UNSUPPORTED, "vince_trunk_forward: staged input needs the packed stem layout");
    } else if (!stem_packed()) {
        if (jig_h > 0)
            RC(vince_jigsaw_nchw_to_nhwc(c.dtype, input, at(workspace, t->off_x0), N / 9, 3, jig_h, jig_w, t->cfg.H, t->cfg.W,
                                         t->Cp, stream));
        else
            RC(vince_input_nchw_to_nhwc(c.dtype, input, perm, at(workspace, t->off_x0), N, 3, t->cfg.H, t->cfg.W, t->Cp, stream));
    } else if (jig_h > 0) {
        VINCE_CHECK_ARG(N % 9 == 0, VINCE_E_SHAPE, "vince_trunk_forward: jigsaw needs N multiple of 9");
        RC(vince_jigsaw_nchw_to_rows(c.dtype, input, at(workspace, t->off_x0), N / 9, 3, jig_h, jig_w, t->cfg.H, t->cfg.W,
                                     t->sWp, STEM_LEFT, stream));
    } else {
        RC(vince_input_nchw_to_rows(c.dtype, input, perm, at(workspace, t->off_x0), N, 3, t->cfg.H, t->cfg.W, t->sWp, STEM_LEFT,
                                    stream));
    }
    // Mixed mode "x3f" (vince_trunk_set_shadow): a grad-enabled train-mode forward of this fp32-tensor handle also leaves everything the
    // backward reads as bfloat16 in the twin's workspace -- raw convolution outputs and activations from the epilogues / passes that write
    // them (one extra 2-byte store per element), the stem input and the pool output by a cast, ReLU masks and pool argmax bytes straight
    // into the twin (this handle's own backward is not going to run), the BatchNorm constants by a copy at the end.
    vince_trunk* const S = (save && t->shadow) ? t->shadow : nullptr;
    void* const sw = t->shadow_ws;
    if (S) {
        RC(stem_join(S, stream));      // (the twin's deferred stem weight gradient may still be reading ITS workspace)
        VINCE_CHECK_ARG(train_bn && stem_packed(), VINCE_E_UNSUPPORTED,
                        "vince_trunk_forward: the bf16 shadow needs a train-mode forward and the packed stem layout");
        RC(vince_cast_f32_to_bf16((const float*)at(workspace, t->off_x0), at(sw, S->off_x0), (size_t)N * t->cfg.H * t->sWp * STEM_CS, stream));
    }
    const vince_conv_desc sd = stem_desc(t);
    // stem: conv 7x7/s2 -> BN -> ReLU -> maxpool 3x3/s2 (resnet.py:170-173); BN-apply + ReLU are fused into the pool
    RC(conv_bn_fwd(c, t->stem, t->stem_bn, t->off_x0, t->off_ystem, bn_running, bn_nbt, train_bn, true, &sd, S ? at(sw, S->off_ystem) : nullptr));
    RC(vince_stem_pool_fwd(c.dtype, at(workspace, t->off_ystem), c.consts(t->stem_bn, 0), c.consts(t->stem_bn, 1),
                           at(workspace, t->off_p0), (uint8_t*)(S ? at(sw, S->off_amax) : at(workspace, t->off_amax)), N, t->sH, t->sW, 64, stream));
    if (S) {
        RC(vince_cast_f32_to_bf16((const float*)at(workspace, t->off_p0), at(sw, S->off_p0), (size_t)N * t->pH * t->pW * 64, stream));
        RC(copy_consts(c, t->stem_bn, S, sw, S->stem_bn));
    }
    // The forward downsample conv on its own stream is OPT-IN (VINCE_DS_STREAM_FWD=1): worth 0.1 ms when it happens to share a
    // hardware queue with another stream (GPU_MAX_HW_QUEUES=4, the default), but +4 ms when every stream gets its own queue
    // (two overlapped encoders x two streams each thrash) -- the mapping depends on stream creation order, so it is not relied on.
    static const int ds_fwd_mode = (vince_knob("ds_stream", 1) == 0) ? 0
                                   : (vince_knob("ds_stream_fwd", 0));
    // mode 2: only a handle that already owns a downsample stream from an earlier backward (the query encoder's) uses it in
    // forward too -- no stream is created for it, the key encoder stays inline
    const bool ds_side = (ds_fwd_mode == 1 || (ds_fwd_mode == 2 && save && t->ds_stream)) && !vince_profile_enabled() &&
                         vince_side_stream_budget() >= 2 && !(save && t->shadow);
    if (ds_side && !t->ds_stream) {
        RC(shared_stream(g_ds_stream, false, 0, &t->ds_stream));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_start, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_dy, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_wg, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_done, hipEventDisableTiming));
    }
    // Gram-statistics residual join (no-grad train-mode forwards: the key encoder, forward + InfoNCE): for a bottleneck's
    // conv3 = 1x1 with a short reduction, the batch statistics of bn3 follow from the Gram matrix of conv3's INPUT
    // (vince_bn_gram_finalize), so they are known BEFORE conv3 runs and its epilogue applies bn3 + identity + ReLU in place on
    // the identity tensor -- y3 is neither written nor re-read and conv3 carries no statistics epilogue (28 -> 21 tensor
    // passes per block).  Backward needs y3, so grad-enabled forwards keep the separate passes.  VINCE_GRAM_JOIN=0: off.
    static const bool gram_env = (vince_knob("gram_join", 1) != 0);
    const bool gram_nograd = gram_env && train_bn && !save && !ds_side;
    // Grad-enabled forwards CAN take the same route where the streaming kernel applies (bf16, K = 64 / 128): it writes the block
    // output AND what backward reads -- conv3's raw output and the ReLU mask bytes -- so the join pass, its re-read of y3 and
    // conv3's statistics epilogue go (17 -> 13 tensor passes for conv3 + join).  OPT-IN (VINCE_GRAM_TRAIN=1): measured neutral in
    // the full step (27.21 vs 27.28 ms: the query forward overlaps the key encoder's, both HBM-bound) while the fp32 atomics of
    // the Gram sums make bn3's constants -- and through bf16 rounding the early-layer gradients -- vary from run to run.
    static const bool xjoin_env = (vince_knob("xjoin", 1) != 0);
    // DEFAULT for bf16 training forwards since round 3 (VINCE_BN3_ALGEBRA, alg_block): the same route WITHOUT storing conv3's output
    // -- backward no longer reads it (csrc/bn_algebra.hip) -- 17 -> 9 tensor passes for conv3 + join.
    const bool alg_fwd = alg_env() && gram_env && xjoin_env && train_bn && save && !ds_side && c.dtype == VINCE_BF16;
    const bool gram_train = gram_env && xjoin_env && (alg_fwd || (vince_knob_live("gram_train", 0) == 1)) &&
                            train_bn && save && !ds_side && c.dtype == VINCE_BF16;
    if (save) t->fwd_alg = alg_fwd;
    // Mixed mode (shadow S): the grad-enabled forward takes the no-grad Gram route too -- conv3 + bn3 + join in one launch, IN PLACE on the
    // fp32 identity (nothing reads it again: the twin holds its bf16 copy) -- with the twin's copies (conv3's output centred, the block
    // output, the ReLU bits) written by that epilogue.  `gram_shadow=0`: the separate passes (cross-check switch).
    const bool gram_shadow = gram_env && train_bn && save && S != nullptr && !ds_side && vince_knob_live("gram_shadow", 1) != 0;
    // ... and where the twin's backward can take the BatchNorm-backward algebra (alg_block: K = 64 / 128) it gets the Gram sums instead of
    // a copy of conv3's output.  `x3f_alg=0`: centred copies everywhere (cross-check switch).
    const bool twin_alg = gram_shadow && alg_env() && vince_knob_live("x3f_alg", 1) != 0;
    const bool gram_on = gram_nograd || gram_train || gram_shadow;
    const bool gram_fused = vince_knob_live("gram_fused", 1) != 0;
    if (gram_on && t->gram_bytes)
        RC(vince_zero_async(at(workspace, t->off_gram), t->gram_bytes, stream));
    size_t cur = t->off_p0;   // where the running block input lives (gram blocks update it in place, so b.x_in may be stale)
    // `xjoin_next`: a layer1 join also runs the NEXT block's conv1 (256 -> 64) on the block output while it is in LDS
    // (vince_conv_expand_join_next: the 411 MB re-read of that launch goes); 0 = every conv1 as its own launch (cross-check switch)
    const bool next_env = vince_knob_live("xjoin_next", 1) != 0;
    bool conv1_done = false;  // block bi's conv1 (output + statistics) came out of block bi-1's join
    for (size_t bi = 0; bi < t->blocks.size(); ++bi) {
        const Blk& b = t->blocks[bi];
        const Blk* const sb = S ? &S->blocks[bi] : nullptr;      // the twin's block: where the bf16 copies go
        const size_t x_in = cur;
        size_t in = x_in;
        const bool skip_conv1 = conv1_done;
        conv1_done = false;
        const bool xj_ok = xjoin_env && c.dtype == VINCE_BF16 && b.nconv == 3 && (b.c[2].Ci == 64 || b.c[2].Ci == 128) &&
                           b.c[2].Co % 256 == 0 &&
                           (unsigned long long)t->cfg.N * b.c[2].Hi * b.c[2].Wi * b.c[2].Ci * 2 < 0x7ff00000ull;   // 31-bit descriptor offsets
        const bool gram_blk = b.gram != NONE && (((gram_nograd || gram_shadow) && bi + 1 < t->blocks.size()) || (gram_train && xj_ok));
        if (b.has_ds && ds_side) {
            Ctx cd = c;
            cd.stream = (void*)t->ds_stream;
            VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_start, (hipStream_t)stream));
            VINCE_CHECK_HIP(hipStreamWaitEvent(t->ds_stream, t->ev_ds_start, 0));
            RC(conv_bn_fwd(cd, b.cd, b.bd, x_in, b.yd, bn_running, bn_nbt, train_bn, true));
            VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_done, t->ds_stream));
        }
        const int nplain = gram_blk ? b.nconv - 1 : b.nconv;   // convs that run with their own statistics epilogue
        bool gram_done = false;
        // Split-half forwards whose activations no fp32 reader needs afterwards (no-grad forwards; grad-enabled ones with a bf16 twin):
        // bn1's output -- read by the block's 3x3 and by nothing else -- is written as stored IEEE-half pairs and the 3x3 multiplies the
        // pairs as they are (VINCE_EPI_IN_HALF_PAIRS): the in-register split is 15-17 % of a split-half 3x3.  `x3_half_pairs=0`: off.
        const bool hp = t->cf == VINCE_F32X3H && train_bn && (!save || S != nullptr) && b.nconv >= 2 && b.c[1].k == 3 && b.c[0].Co % 16 == 0 &&
                        (unsigned long long)N * b.c[0].Ho * b.c[0].Wo * b.c[0].Co * 4 < 0x7ff00000ull && vince_knob_live("x3_half_pairs", 1) != 0;
        for (int ci = 0; ci < nplain; ++ci) {
            if (!(ci == 0 && skip_conv1))
                RC(conv_bn_fwd(c, b.c[ci], b.b[ci], in, b.y[ci], bn_running, bn_nbt, train_bn, false, nullptr, nullptr, hp && ci == 1));
            if (ci < b.nconv - 1) {
                // (the pass that writes conv3's input also sums it per channel when the Gram path follows -- and, for the bf16
                // K = 64 / 128 blocks, multiplies what it writes into the Gram matrix itself: csrc/bn_gram.hip, `gram_fused=0` restores the
                // weight-gradient launch over the stored tensor)
                double* osum = (gram_blk && ci == b.nconv - 2) ? (double*)at(workspace, b.colsum) : nullptr;
                if (osum && gram_fused && c.dtype == VINCE_BF16 && (b.c[ci].Co == 64 || b.c[ci].Co == 128) && t->wg_scratch_bytes) {
                    RC(bn_apply_gram_fwd(c, b.c[ci], b.b[ci], b.y[ci], at(workspace, b.a[ci]), bn_running, bn_nbt, osum,
                                         (float*)at(workspace, b.gram), at(workspace, t->off_wg_scratch[0]), t->wg_scratch_bytes));
                    gram_done = true;
                } else
                RC(bn_apply_fwd(c, b.c[ci], b.b[ci], b.y[ci], nullptr, nullptr, nullptr, at(workspace, b.a[ci]), nullptr,
                                bn_running, bn_nbt, train_bn, osum, sb ? at(sw, sb->a[ci]) : nullptr, nullptr,
                                sb ? at(sw, sb->y[ci]) : nullptr, sb ? twin_consts(S, sw, sb->b[ci]) : nullptr, hp && ci == 0 && !osum));
                in = b.a[ci];
            }
        }
        const int L = b.nconv - 1;
        uint8_t* zmask = (uint8_t*)at(workspace, b.zmask);
        if (gram_blk) {
            const ConvL& cv = b.c[L];
            const BnL& bn = b.b[L];
            const int64_t rows = (int64_t)N * cv.Ho * cv.Wo;
            // Gram matrix of conv3's input through the weight-gradient kernel (in = dy = a): sum over pixels of a a^T
            vince_conv_desc dg = fwd_desc(t, cv);
            dg.Co = cv.Ci;
            {
                // (split-half modes: the Gram matrix as split-half products too -- bfloat16 hi + lo parts, three products, 2^-17 per term --
                // instead of exact fp32 MFMAs at a fifth of the rate: 24 launches, 2.1 ms per step.  `gram_x3=0`: exact fp32, as until round 6.)
                const int gdt = (t->cf == VINCE_F32X3H && vince_knob_live("gram_x3", 1) != 0) ? VINCE_F32X3B : c.dtype;
                if (!gram_done)
                RC(wgrad_launch(t, workspace, gdt, dg, at(workspace, in), at(workspace, in), (float*)at(workspace, b.gram), cv.Ci, 0, stream));
            }
            // (split-half mode: the cache holds hi / lo half pairs, the finalize reads conv3's fp32 master weights -- [Co][1][1][Ci], the
            // same [Co][K] rows -- instead)
            const void* w3 = t->cf == VINCE_F32X3H ? (const void*)params[cv.param] : (const void*)at((void*)wcache, cv.wk);
            RC(vince_bn_gram_finalize(c.dtype, (const float*)at(workspace, b.gram), (const double*)at(workspace, b.colsum), GRAM_R,
                                      rows, w3, cv.Ci, cv.Co, params[bn.gamma], params[bn.beta],
                                      bn_running[2 * bn.index], bn_running[2 * bn.index + 1], bn_nbt ? bn_nbt[bn.index] : nullptr,
                                      0.1f, 1e-5f, c.consts(bn, 0), c.consts(bn, 1), c.consts(bn, 2), c.consts(bn, 3), stream));
            vince_conv_epi e;
            memset(&e, 0, sizeof(e));
            e.flags = VINCE_EPI_ACCUMULATE | VINCE_EPI_RELU;
            e.out_scale = c.consts(bn, 0);
            e.bias = c.consts(bn, 1);
            size_t idn = x_in, out = x_in;     // no-grad, identity blocks: the join lands on the block input, in place
            if (b.has_ds) {                    // stage entry: on the downsample conv's raw output, read through its BatchNorm affine
                RC(conv_bn_fwd(c, b.cd, b.bd, x_in, b.yd, bn_running, bn_nbt, train_bn, true, nullptr, sb ? at(sw, sb->yd) : nullptr));
                if (sb) RC(copy_consts(c, b.bd, S, sw, sb->bd));
                e.id_scale = c.consts(b.bd, 0);
                e.id_shift = c.consts(b.bd, 1);
                idn = out = b.yd;
            }
            if (save && !gram_shadow) out = b.z;   // backward reads the identity tensors again: nothing in place
            if (gram_shadow) {                 // the twin's view of this block's tail
                e.out2 = at(sw, sb->z);
                e.mask2 = (uint8_t*)at(sw, sb->zmask);
                if (twin_alg && alg_block(S, bi)) {
                    // its backward takes the BatchNorm-backward algebra (csrc/bn_algebra.hip): no copy of conv3's output at all -- the Gram
                    // matrix and column sums of conv3's input (adjacent in both workspaces) and bn3's constants as they are
                    VINCE_CHECK_ARG(sb->gram != NONE && sb->colsum == sb->gram + (b.colsum - b.gram), VINCE_E_ARG,
                                    "vince_trunk_forward: the twin's Gram scratch is laid out differently");
                    VINCE_CHECK_HIP(hipMemcpyAsync(at(sw, sb->gram), at(workspace, b.gram),
                                                   (b.colsum - b.gram) + (size_t)GRAM_R * cv.Ci * sizeof(double), hipMemcpyDeviceToDevice,
                                                   (hipStream_t)stream));
                    RC(copy_consts(c, bn, S, sw, sb->b[L]));
                } else {                       // the separate BatchNorm-backward passes: conv3's output centred, constants to match
                    e.raw2 = at(sw, sb->y[L]);
                    e.raw2_mean = c.consts(bn, 2);
                    hipLaunchKernelGGL(centre_consts_kernel, dim3((bn.C + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                       (const float*)c.consts(bn, 0), twin_consts(S, sw, sb->b[L]), bn.C);
                    VINCE_CHECK_LAUNCH();
                }
            }
            // bf16 at K = 64 / 128: the persistent streaming kernel (csrc/conv_xjoin.hip); otherwise the implicit-GEMM kernel's join
            // epilogue (fp32, or VINCE_XJOIN=0 as a cross-check; no-grad forwards only)
            const Blk* nb = bi + 1 < t->blocks.size() ? &t->blocks[bi + 1] : nullptr;
            // (256 -> 64: layer1's identity blocks; 256 -> 128: the first block of layer2 behind a layer1 block with a plain identity)
            const bool fuse_next = next_env && xj_ok && train_bn && cv.Ci == 64 && cv.Co == 256 && nb && nb->nconv == 3 &&
                                   nb->c[0].k == 1 && nb->c[0].stride == 1 && nb->c[0].Ci == 256 &&
                                   ((nb->c[0].Co == 64 && !nb->has_ds) || (nb->c[0].Co == 128 && !e.id_scale && next_env128()));
            if (fuse_next) {
                RC(vince_conv_expand_join_next(c.dtype, at(workspace, in), at((void*)wcache, cv.wk), rows, cv.Ci, cv.Co, e.out_scale, e.bias,
                                               at(workspace, idn), e.id_scale, e.id_shift, at(workspace, out),
                                               (save && !(alg_fwd && alg_block(t, bi))) ? at(workspace, b.y[L]) : nullptr,
                                               save ? zmask : nullptr, 1, at((void*)wcache, nb->c[0].wk), nb->c[0].Co,
                                               at(workspace, nb->y[0]), c.stats(nb->b[0]), nb->b[0].R, nullptr, 0, stream));
                conv1_done = true;
            } else if (xj_ok) {
                RC(vince_conv_expand_join(c.dtype, at(workspace, in), at((void*)wcache, cv.wk), rows, cv.Ci, cv.Co, e.out_scale, e.bias,
                                          at(workspace, idn), e.id_scale, e.id_shift, at(workspace, out),
                                          (save && !(alg_fwd && alg_block(t, bi))) ? at(workspace, b.y[L]) : nullptr,
                                          save ? zmask : nullptr, 1, stream));
            } else {
                const vince_conv_desc d3 = fwd_desc(t, cv);
                RC(vince_conv_igemm(&d3, t->cf, at(workspace, in), at((void*)wcache, cv.wk), at(workspace, out), &e, stream));
            }
            cur = out;
            continue;
        }
        // (with a shadow the ReLU bits go to the twin alone, in ITS format -- one byte per 8 channels)
        void* const z2 = sb ? at(sw, sb->z) : nullptr;
        uint8_t* const zm2 = sb ? (uint8_t*)at(sw, sb->zmask) : nullptr;
        uint8_t* const zm1 = sb ? nullptr : zmask;
        void* const y2L = sb ? at(sw, sb->y[L]) : nullptr;       // the block's last BatchNorm: centred shadow of its input + the twin's constants
        float* const c2L = sb ? twin_consts(S, sw, sb->b[L]) : nullptr;
        if (b.has_ds) {   // the downsample BatchNorm enters the join as an affine of its conv output: finalised on its own
            if (ds_side) VINCE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, t->ev_ds_done, 0));
            else RC(conv_bn_fwd(c, b.cd, b.bd, x_in, b.yd, bn_running, bn_nbt, train_bn, true, nullptr, sb ? at(sw, sb->yd) : nullptr));
            if (sb) RC(copy_consts(c, b.bd, S, sw, sb->bd));     // (the downsample BatchNorm: raw bf16 shadow from its epilogue, constants as they are)
            RC(bn_apply_fwd(c, b.c[L], b.b[L], b.y[L], at(workspace, b.yd), c.consts(b.bd, 0), c.consts(b.bd, 1),
                            at(workspace, b.z), zm1, bn_running, bn_nbt, train_bn, nullptr, z2, zm2, y2L, c2L));
        } else {
            RC(bn_apply_fwd(c, b.c[L], b.b[L], b.y[L], at(workspace, x_in), nullptr, nullptr, at(workspace, b.z), zm1,
                            bn_running, bn_nbt, train_bn, nullptr, z2, zm2, y2L, c2L));
        }
        cur = b.z;
    }
    RC(vince_avgpool_fwd(c.dtype, at(workspace, t->blocks.back().z), pooled, N, t->outH * t->outW, t->outC, stream));
    if (S) S->fwd_alg = twin_alg;  // (elsewhere its backward takes the separate BatchNorm-backward passes over conv3's centred outputs)
    return VINCE_OK;
}

extern "C" int vince_trunk_backward(vince_trunk_t t, const float* const* params, const void* wcache, void* workspace,
                                    const float* dpooled, float* const* grads, const int32_t* event_blocks,
                                    void* const* events, int32_t n_events, void* stream) {
    VINCE_CHECK_ARG(t && params && wcache && workspace && dpooled && grads, VINCE_E_ARG, "vince_trunk_backward: null pointer");
    Ctx c{t, params, wcache, workspace, stream, t->sdtype};
    const int N = t->cfg.N;
    RC(stem_join(t, stream));
    RC(vince_zero_async(at(workspace, t->off_sums), t->n_stats_doubles * sizeof(double), stream));
    if (t->fwd_alg && t->algR_bytes) RC(vince_zero_async(at(workspace, t->off_algR), t->algR_bytes, stream));
    // Weight gradients run on a side stream: wgrad(layer) only needs dY(layer) and the saved activation, and nothing but
    // the optimiser needs its result, so it overlaps the BatchNorm-backward / dgrad chain of the layers below (compute-
    // bound MFMA work next to HBM-bound streams).  dY lives in a 3-slot ring; a slot is rewritten only after the wgrad
    // that read it has finished (ev_wg), and a wgrad starts when its dY is complete (ev_dy).
    // (per-kernel event timing wants kernels to run alone: overlap is off while vince_profile_enable(1) is in effect)
    static const bool overlap_env = (vince_knob("wgrad_stream", 1) != 0);
    const bool overlap = overlap_env && !vince_profile_enabled() && vince_side_stream_budget() >= 1;
    hipStream_t main_s = (hipStream_t)stream;
    if (overlap && !t->side) {
        // The weight gradients are off the critical path (the chain of dgrad / BatchNorm launches on the caller's stream):
        // their stream gets the LOWEST priority so that a 1000-workgroup wgrad never delays the next dgrad's start
        // (-0.1 ms/step, 4 of 4 paired runs).  VINCE_SIDE_PRIO: 1 lowest (default), 0 same as the caller's, -1 highest.
        static const int side_prio = VINCE_MEASURE_KNOB("side_prio", 1);
        RC(shared_stream(g_side_stream, side_prio != 0, side_prio, &t->side));
        for (int i = 0; i < t->ndy; ++i) {
            VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_dy[i], hipEventDisableTiming));
            VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_wg[i], hipEventDisableTiming));
        }
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_alg, hipEventDisableTiming));
    }
    // the downsample branch of a stage-entry block only meets the main chain again at the block-input gradient: it runs on a
    // third stream (VINCE_DS_STREAM=0: inline on the main stream)
    static const bool ds_env = (vince_knob("ds_stream", 1) != 0) &&
                               (vince_knob("ds_stream_bwd", 1) != 0);
    const bool ds_overlap = overlap && ds_env && vince_side_stream_budget() >= 2;
    if (ds_overlap && !t->ds_stream) {
        RC(shared_stream(g_ds_stream, false, 0, &t->ds_stream));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_start, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_dy, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_wg, hipEventDisableTiming));
        VINCE_CHECK_HIP(hipEventCreateWithFlags(&t->ev_ds_done, hipEventDisableTiming));
    }
    bool ds_wg_pending = false, ds_done_pending = false;
    for (int i = 0; i < t->ndy; ++i) t->wg_pending[i] = false;
    int slot = 0;
    void* DY = nullptr;
    auto next_dy = [&]() -> int {      // claim the next ring slot for writing on the main stream
        slot = (slot + 1) % t->ndy;
        if (overlap && t->wg_pending[slot]) {
            VINCE_CHECK_HIP(hipStreamWaitEvent(main_s, t->ev_wg[slot], 0));
            t->wg_pending[slot] = false;
        }
        DY = at(workspace, t->off_dy[slot]);
        return VINCE_OK;
    };
    auto wgrad_async = [&](const vince_conv_desc& d, const void* in, float* dw, int ci_dw) -> int {
        if (!overlap) return wgrad_launch(t, workspace, t->cb, d, in, DY, dw, ci_dw, 0, stream);
        VINCE_CHECK_HIP(hipEventRecord(t->ev_dy[slot], main_s));
        VINCE_CHECK_HIP(hipStreamWaitEvent(t->side, t->ev_dy[slot], 0));
        RC(wgrad_launch(t, workspace, t->cb, d, in, DY, dw, ci_dw, 1, (void*)t->side));
        VINCE_CHECK_HIP(hipEventRecord(t->ev_wg[slot], t->side));
        t->wg_pending[slot] = true;
        return VINCE_OK;
    };
    auto wgrad_layer = [&](const ConvL& cv, const void* in) -> int {
        return wgrad_async(fwd_desc(t, cv), in, grads[cv.param], cv.Ci);
    };
    void* Z = at(workspace, t->off_g[0]);
    void* DA = at(workspace, t->off_g[1]);
    void* DX = at(workspace, t->off_g[2]);
    RC(vince_avgpool_bwd(c.dtype, dpooled, Z, N, t->outH * t->outW, t->outC, stream));
    // BatchNorm-backward reductions ride in the epilogue of the dgrad that produces their input gradient
    // (VINCE_FUSE_BNRED=0 runs them as separate passes: measurement aid)
    static const bool fuse_red = (VINCE_MEASURE_KNOB("fuse_bnred", 1) != 0);
    static const bool wgrad_late = VINCE_MEASURE_KNOB("wgrad_late", 0) != 0;
    bool last_reduced = false;   // was the last-BN reduction of the current block done by the block above it?
    for (int bi = (int)t->blocks.size() - 1; bi >= 0; --bi) {
        const Blk& b = t->blocks[bi];
        const int L = b.nconv - 1;
        const ConvL& last = b.c[L];
        const int64_t rows_out = (int64_t)N * last.Ho * last.Wo;
        const uint8_t* zbits = (const uint8_t*)at(workspace, b.zmask);
        const void* x_in = at(workspace, b.x_in);
        // BatchNorm-backward algebra (csrc/bn_algebra.hip): for these blocks Z already holds g = dz * (z > 0) -- the producer's
        // epilogue gated it with this block's ReLU bits -- and c.sums(bn_L) its per-channel sums; bn_L's backward and conv_L's
        // gradients then need neither y_L nor a dY tensor.  alg_lo: the block BELOW is one, so THIS block's input-gradient
        // launch gates and sums for it.
        const bool alg = t->fwd_alg && alg_block(t, (size_t)bi);
        const bool alg_lo = bi > 0 && t->fwd_alg && alg_block(t, (size_t)bi - 1);
        int ci_top = L;
        if (alg) {
            if (b.has_ds) {
                // downsample branch: its BatchNorm consumes the same g (already gated: no mask), with its own reduction pass
                if (ds_overlap) {
                    void* DYD = at(workspace, t->off_dyd);
                    Ctx cd = c;
                    cd.stream = (void*)t->ds_stream;
                    VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_start, main_s));
                    VINCE_CHECK_HIP(hipStreamWaitEvent(t->ds_stream, t->ev_ds_start, 0));
                    if (ds_wg_pending) VINCE_CHECK_HIP(hipStreamWaitEvent(t->ds_stream, t->ev_ds_wg, 0));   // DYD still being read
                    RC(bn_bwd(cd, b.bd, Z, nullptr, false, b.yd, rows_out, DYD, nullptr, grads, false));
                    VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_dy, t->ds_stream));
                    VINCE_CHECK_HIP(hipStreamWaitEvent(t->side, t->ev_ds_dy, 0));
                    {
                        const vince_conv_desc dd = fwd_desc(t, b.cd);
                        RC(wgrad_launch(t, workspace, t->cb, dd, x_in, DYD, grads[b.cd.param], b.cd.Ci, 1, (void*)t->side));
                    }
                    VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_wg, t->side));
                    ds_wg_pending = true;
                    RC(dgrad(cd, b.cd, DYD, DX, false));
                    VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_done, t->ds_stream));
                    ds_done_pending = true;
                } else {
                    RC(next_dy());
                    RC(bn_bwd(c, b.bd, Z, nullptr, false, b.yd, rows_out, DY, nullptr, grads, false));
                    RC(wgrad_layer(b.cd, x_in));
                    RC(dgrad(c, b.cd, DY, DX, false));
                }
            }
            const ConvL& cv = b.c[L];
            const BnL& bn = b.b[L];
            const void* a_in = at(workspace, b.a[L - 1]);
            // A twin (mixed mode: the forward multiplied with the fp32 masters) derives the coefficients from the MASTER weights and takes the
            // input gradient's matrices in bf16 hi + lo parts -- what keeps the masked sums the BatchNorm below reduces at the separate
            // passes' accuracy (csrc/bn_algebra.hip; `alg_split=0`: single bf16 matrices from the bf16 cache, as the bf16 mode has them)
            // (`alg_split`: 2 = nq in two parts, the default -- its rounding residue multiplies a, which is zero exactly where bn2's mask is;
            // 1 = wd as well, +0.25 ms: 3x closer on synthetic data, tools/alg_op_probe.py, no measurable difference on G9 / G12; 0 = neither)
            const int split_mode = t->is_twin ? (int)vince_knob_live("alg_split", 2) : 0;
            const bool split = split_mode != 0, split_wd = split_mode == 1;
            const void* wk = split ? (const void*)params[cv.param] : at((void*)wcache, cv.wk);      // ([Co][1][1][Ci] = the same [Co][K] rows)
            const int wdt = split ? VINCE_F32 : VINCE_BF16, taps = split_wd ? 3 : 2;
            const AlgPtrs ap = alg_ptrs(workspace, b);
            unsigned char* const w2b = (unsigned char*)ap.w2;
            // R = g^T a into this block's scratch (on this stream: the algebra below needs it before the dgrad); the finished weight
            // gradient is then ADDED into the gradient buffer like every other one
            float* const R = (float*)at(workspace, b.algR);
            {
                const vince_conv_desc dw = fwd_desc(t, cv);
                RC(wgrad_launch(t, workspace, t->cb, dw, a_in, Z, R, cv.Ci, 0, stream));
            }
            RC(vince_bn3_bwd_prepare(R, wk, c.sums(bn), bn.R, c.consts(bn, 2), c.consts(bn, 3), params[bn.gamma], rows_out,
                                     cv.Co, cv.Ci, ap.coef, ap.w2, taps * cv.Co, w2b + (size_t)(taps - 1) * cv.Co * 2, taps * cv.Co, ap.nr,
                                     grads[bn.gamma], grads[bn.beta], (const double*)at(workspace, b.colsum), GRAM_R, wdt,
                                     split_wd ? w2b + (size_t)cv.Co * 2 : nullptr,
                                     split ? w2b + ((size_t)(taps - 1) * cv.Co + cv.Ci) * 2 : nullptr, stream));
            // the finished weight gradient is nobody's input but the optimiser's: on the weight-gradient stream, behind the coefficients
            // (ev_alg), off the chain of launches the input gradient below waits for
            if (overlap) {
                VINCE_CHECK_HIP(hipEventRecord(t->ev_alg, main_s));
                VINCE_CHECK_HIP(hipStreamWaitEvent(t->side, t->ev_alg, 0));
            }
            RC(vince_bn3_bwd_finish_dw(R, grads[cv.param], wk, (const float*)at(workspace, b.gram), (const double*)at(workspace, b.colsum),
                                       GRAM_R, ap.coef, ap.coef + 4 * (size_t)cv.Co, c.consts(bn, 3), cv.Co, cv.Ci, wdt,
                                       overlap ? (void*)t->side : stream));
            // da = (W^T diag(s)) g + nq a + nr in ONE launch: the reduction runs over g's 4w channels (tap 0) and then over a's w
            // channels (tap 1 = vince_conv_epi.in2), with the fused reduction of the BatchNorm below as the plain dgrad has it
            {
                vince_conv_desc ds1[4];
                const int n1 = dgrad_descs(t, cv, ds1);
                if (n1 != 1) {
                    vince_set_error("vince_trunk_backward: the BatchNorm-backward algebra expects a stride-1 1x1 convolution");
                    return VINCE_E_UNSUPPORTED;
                }
                vince_conv_desc dq = ds1[0];
                dq.TA = 1; dq.TB = taps; dq.dh0 = dq.dw0 = 0; dq.dhs = 1; dq.dws = 0;
                dq.wt0 = 0; dq.wta = 0; dq.wtb = 1; dq.WT = taps;
                vince_conv_epi e1;
                memset(&e1, 0, sizeof(e1));
                e1.bias = ap.nr;
                e1.in2 = a_in;
                e1.in2_channels = cv.Ci;
                e1.in2_repeat = split ? 2 : 0;
                if (fuse_red) e1.bnred = bn_reduce_of(c, b.b[L - 1], nullptr, true, b.y[L - 1]);
                e1.replicas = b.b[L - 1].R;
                RC(vince_conv_igemm(&dq, c.dtype, Z, ap.w2, DA, &e1, stream));
            }
            {
                const int64_t rows = (int64_t)N * b.c[L - 1].Ho * b.c[L - 1].Wo;
                RC(next_dy());
                RC(bn_bwd(c, b.b[L - 1], DA, nullptr, true, b.y[L - 1], rows, DY, nullptr, grads, fuse_red));
            }
            ci_top = L - 1;
        } else
        // z = relu(bn_L(y_L) + identity): g = dz * (z > 0) is the gradient of both addends
        if (b.has_ds) {
            // bn_L's backward apply also accumulates the downsample BatchNorm's reduction (same g, its own y): one pass
            // over dz instead of two.  Its dY keeps ring slot A for the main chain below; the downsample branch (apply,
            // wgrad, dgrad -> DX) runs in the next slot first.  (VINCE_FUSE_DS_REDUCE=0: separate reduce pass)
            static const bool fuse_ds = (VINCE_MEASURE_KNOB("fuse_ds_reduce", 1) != 0);
            vince_bn_reduce2 r2;
            r2.y = at(workspace, b.yd);
            r2.mean = c.consts(b.bd, 2);
            r2.invstd = c.consts(b.bd, 3);
            r2.sums = c.sums(b.bd);
            r2.replicas = b.bd.R;
            RC(next_dy());
            RC(bn_bwd(c, b.b[L], Z, zbits, false, b.y[L], rows_out, DY, nullptr, grads, last_reduced, fuse_ds ? &r2 : nullptr));
            if (ds_overlap) {
                void* DYD = at(workspace, t->off_dyd);
                Ctx cd = c;
                cd.stream = (void*)t->ds_stream;
                VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_start, main_s));
                VINCE_CHECK_HIP(hipStreamWaitEvent(t->ds_stream, t->ev_ds_start, 0));
                if (ds_wg_pending) VINCE_CHECK_HIP(hipStreamWaitEvent(t->ds_stream, t->ev_ds_wg, 0));   // DYD still being read
                RC(bn_bwd(cd, b.bd, Z, zbits, false, b.yd, rows_out, DYD, nullptr, grads, fuse_ds));
                VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_dy, t->ds_stream));
                VINCE_CHECK_HIP(hipStreamWaitEvent(t->side, t->ev_ds_dy, 0));
                {
                    const vince_conv_desc dd = fwd_desc(t, b.cd);
                    RC(wgrad_launch(t, workspace, t->cb, dd, x_in, DYD, grads[b.cd.param], b.cd.Ci, 1, (void*)t->side));
                }
                VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_wg, t->side));
                ds_wg_pending = true;
                RC(dgrad(cd, b.cd, DYD, DX, false));
                VINCE_CHECK_HIP(hipEventRecord(t->ev_ds_done, t->ds_stream));
                ds_done_pending = true;
            } else {
                void* const dy_main = DY;
                const int slot_main = slot;
                RC(next_dy());
                RC(bn_bwd(c, b.bd, Z, zbits, false, b.yd, rows_out, DY, nullptr, grads, fuse_ds));
                RC(wgrad_layer(b.cd, x_in));
                RC(dgrad(c, b.cd, DY, DX, false));
                DY = dy_main;
                slot = slot_main;
            }
        } else {
            // identity branch: g = dz * (z > 0) is never materialised -- the block-input dgrad below joins it in place
            RC(next_dy());
            RC(bn_bwd(c, b.b[L], Z, zbits, false, b.y[L], rows_out, DY, nullptr, grads, last_reduced));
        }
        for (int ci = ci_top; ci >= 0; --ci) {
            const void* in_act = ci == 0 ? x_in : at(workspace, b.a[ci - 1]);
            // wgrad_late: the weight gradient of a layer starts after that layer's dgrad instead of next to it, i.e. it
            // runs beside the (HBM-bound) BatchNorm backward of the layer below (VINCE_WGRAD_LATE, measurement knob)
            if (!wgrad_late) RC(wgrad_layer(b.c[ci], in_act));
            if (ci > 0) {
                vince_bn_reduce br = bn_reduce_of(c, b.b[ci - 1], nullptr, true, b.y[ci - 1]);
                RC(dgrad(c, b.c[ci], DY, DA, false, nullptr, fuse_red ? &br : nullptr, b.b[ci - 1].R));
                if (wgrad_late) RC(wgrad_layer(b.c[ci], in_act));
                const int64_t rows = (int64_t)N * b.c[ci - 1].Ho * b.c[ci - 1].Wo;
                RC(next_dy());
                RC(bn_bwd(c, b.b[ci - 1], DA, nullptr, true, b.y[ci - 1], rows, DY, nullptr, grads, fuse_red));
            } else {
                // block-input gradient; it is the dz of the block below, whose last BatchNorm's reduction is fused here
                vince_bn_reduce br;
                const bool fuse = fuse_red && bi > 0 && !alg_lo;
                int rr = 0;
                const uint8_t* lo_mask = nullptr;
                double* lo_sums = nullptr;
                if (fuse) {
                    const Blk& lo = t->blocks[bi - 1];
                    br = bn_reduce_of(c, lo.b[lo.nconv - 1], (const uint8_t*)at(workspace, lo.zmask), false, lo.y[lo.nconv - 1]);
                    rr = lo.b[lo.nconv - 1].R;
                }
                if (alg_lo) {   // the block below runs the algebra: hand it g (gated by ITS ReLU bits) and the sums of g
                    const Blk& lo = t->blocks[bi - 1];
                    lo_mask = (const uint8_t*)at(workspace, lo.zmask);
                    lo_sums = c.sums(lo.b[lo.nconv - 1]);
                    rr = lo.b[lo.nconv - 1].R;
                }
                if (b.has_ds && ds_done_pending) {   // DX holds the downsample branch's gradient once its stream is done
                    VINCE_CHECK_HIP(hipStreamWaitEvent(main_s, t->ev_ds_done, 0));
                    ds_done_pending = false;
                }
                if (b.has_ds) RC(dgrad(c, b.c[0], DY, DX, true, nullptr, fuse ? &br : nullptr, rr, lo_mask, lo_sums));
                else RC(dgrad(c, b.c[0], DY, Z, true, alg ? nullptr : zbits, fuse ? &br : nullptr, rr, lo_mask, lo_sums));   // Z <- dgrad + Z * (z > 0) (Z already gated under the algebra)
                if (wgrad_late) RC(wgrad_layer(b.c[0], in_act));
                last_reduced = fuse;
            }
        }
        if (b.has_ds) std::swap(Z, DX);
        // gradient buckets: every parameter of blocks >= bi is final once BOTH streams have passed this point -- the
        // bucket event is recorded on the side stream behind a join with the main stream
        for (int e = 0; e < n_events; ++e)
            if (event_blocks[e] == bi) {
                if (overlap) {
                    VINCE_CHECK_HIP(hipEventRecord(t->ev_join, main_s));
                    VINCE_CHECK_HIP(hipStreamWaitEvent(t->side, t->ev_join, 0));
                    VINCE_CHECK_HIP(hipEventRecord((hipEvent_t)events[e], t->side));
                } else {
                    VINCE_CHECK_HIP(hipEventRecord((hipEvent_t)events[e], main_s));
                }
                if (t->bucket_cb) t->bucket_cb(e, t->bucket_cb_user);   // e.g. enqueue this bucket's all-reduce behind the event NOW
            }
    }
    // stem: Z = gradient wrt the pooled stem output
    // (max-pool backward gathered on the fly inside the bn1 backward passes: the pre-pool gradient is never materialised)
    RC(next_dy());
    {
        const BnL& sb = t->stem_bn;
        const uint8_t* amax = (const uint8_t*)at(workspace, t->off_amax);
        const void* ys = at(workspace, t->off_ystem);
        static const bool fused_stem = (VINCE_MEASURE_KNOB("fuse_stem_bwd", 1) != 0);
        if (fused_stem) {
            RC(vince_stem_bwd_reduce(c.dtype, Z, amax, ys, c.consts(sb, 2), c.consts(sb, 3), c.sums(sb), N, t->sH, t->sW, 64, stream));
            RC(vince_stem_bwd_apply(c.dtype, Z, amax, ys, c.consts(sb, 2), c.consts(sb, 3), params[sb.gamma], c.sums(sb), DY,
                                    grads[sb.gamma], grads[sb.beta], N, t->sH, t->sW, 64, stream));
        } else {
            RC(vince_stem_pool_bwd(c.dtype, Z, amax, DA, N, t->sH, t->sW, 64, stream));
            RC(bn_bwd(c, sb, DA, nullptr, false, t->off_ystem, (int64_t)N * t->sH * t->sW, DY, nullptr, grads));
        }
    }
    if (t->stem_event && overlap) {
        // Deferred stem join (vince_trunk_set_stem_event): the caller's stream waits for every weight gradient BUT the stem's -- whatever
        // it enqueues next (the optimiser over every other parameter, the key encoder's EMA) runs beside that launch, which is
        // otherwise alone on the machine at the very end of the step (~200 us at ResNet-50, B = 256) -- and the stem's gradient is
        // final once stem_event has passed.  The launch keeps reading the stem input, its dY slot and the scratch after this function
        // has returned: `stem_inflight` makes the next forward / backward / folded forward on this handle wait for the event (stem_join).
        VINCE_CHECK_HIP(hipEventRecord(t->ev_join, t->side));
        VINCE_CHECK_HIP(hipStreamWaitEvent(main_s, t->ev_join, 0));
        for (int i = 0; i < t->ndy; ++i) t->wg_pending[i] = false;
        RC(wgrad_async(stem_desc(t), at(workspace, t->off_x0), grads[t->stem.param], 3));
        VINCE_CHECK_HIP(hipEventRecord(t->stem_event, t->side));
        t->stem_inflight = true;
        return VINCE_OK;
    }
    RC(wgrad_async(stem_desc(t), at(workspace, t->off_x0), grads[t->stem.param], 3));
    if (overlap) {   // the caller's stream continues only after every weight gradient has landed
        VINCE_CHECK_HIP(hipEventRecord(t->ev_join, t->side));
        VINCE_CHECK_HIP(hipStreamWaitEvent(main_s, t->ev_join, 0));
        for (int i = 0; i < t->ndy; ++i) t->wg_pending[i] = false;
    }
    if (t->stem_event) VINCE_CHECK_HIP(hipEventRecord(t->stem_event, main_s));   // (streams serialised: already final here)
    return VINCE_OK;
}
