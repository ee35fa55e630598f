// vkn_api.hip — the C ABI of libvkn.so (include/vkn.h): argument checking, workspace carving and the launch sequence of
// one stage / of the S-stage loop.  Host code only; every kernel lives in vkn_gather / vkn_update / vkn_decode.
#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

namespace {

struct Carver {
    char* base;
    size_t off;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int npt_of(int N) { return (N + 31) / 32 * 32; }

struct StageWs {
    float *part, *cntp, *xraw, *cnt, *xfeat, *params, *inputf, *ig, *ug, *f, *obj1, *qkv, *ao, *obj2, *h, *partial, *t1, *t2,
        *maskfeat, *kb, *kern32, *lq, *lkv, *lobj, *lupd, *lf;
    _Float16 *kfh, *kfl;
    int* status;   // the workspace's status word: its FIRST four bytes (include/vkn.h: vkn_workspace_status)
};

// Optional previous-frame blocks of the video head's LAST stage (knet/video/kernel_update_head.py:192-236).  A "link block" is
//   kv  = KernelUpdator_link(update_feature, prev)   (only when the block's weights carry an updator)
//   out = LN(FFN(LN(cur + MHA_8(q = cur, k = v = kv))))
// and is described by a VknStageWeights whose kernel_update_conv.* / attention_previous.* / link_ffn.* members hold the block's
// weights (everything else NULL).  It covers previous_link = "update_dynamic_cov" | "link_atten" (cur = the stage's incoming kernels:
// the result REPLACES them, :324-372) and previous_type = "ffn" | "update" | "update_obj" (cur = the stage's updated kernels: the
// result is the tracking embedding, :394-476).
struct StageOpts {
    const VknStageWeights* link_pre = nullptr;    // previous_link block, applied to obj_in before the update
    const float* prev_pre = nullptr;              // its previous-frame kernels [B][N][C]
    const VknStageWeights* link_track = nullptr;  // previous_type "update" / "update_obj" block (NULL: the stage's own "ffn" link)
    int track_src = 0;                            // update feature of link_track's updator: 1 = x_feat, 2 = the stage's obj_out
    bool skip_decode = false;                     // stop after the decode kernels (planes / kern32, kb) are written
    bool keep_xfeat = false;                      // materialise x_feat in the workspace: a caller-side link reads it after the stage
    const void* touch_next = nullptr;             // the NEXT stage's prepared weights: warmed by the reduction that ends this stage's
    size_t touch_next_bytes = 0;                  // fused decode -> gather pass (k_gather_reduce)
};

// pre-split (bf16x3) copies of the Linear weights, carved from VknStageWeights.prepared in a fixed order
struct PrepW {
    const void *ft, *ftT, *dyn, *inp, *ig, *ug, *fc, *attn_in, *attn_out, *ffn1, *ffn2, *cls_fc[VKN_MAX_FCS], *fc_cls,
        *mask_fc[VKN_MAX_FCS], *fc_mask, *pa_in, *pa_in_kv, *pa_out, *lffn1, *lffn2;
    // composite weights (feat_transform folded into its consumers, computed once in vkn_prepare_stage_f32):
    //   dynft = W_dyn.W_ft [2C][C], bcnt = W_dyn.b_ft [2C]       -> dynamic_layer(x_feat) straight from the raw gather
    //   dec = W_ft^T.W_fm [C][C], decb = W_ft^T.b_fm [C]          -> decode kernels straight from the last mask_fcs output
    //   dvec = W_fm^T.b_ft [C], kb0 = b_fm.b_ft                   -> decode bias
    const void *dynft, *dec;
    float *dynft32, *bcnt, *dec32, *decb, *dvec, *kb0, *fmT;
    float* chain_consts;   // packed bias / LayerNorm vectors of the persistent chain kernels (vkn_chain.hip), C == 256 only
    // the persistent chain's weights once more as fp16 hi / lo tile images (vkn_chain_h2.hip: the default of the persistent form), in VKN_H2_* order;
    // h2_scale [VKN_H2_COUNT][8]: vkn_pow2_scale_f32's output per matrix ([0] scale, [4] 1 / scale); h2_scratch: its two words
    const void* h2[VKN_H2_COUNT];
    float* h2_scale;
    unsigned* h2_scratch;
};

inline bool has_composites(const VknDims* d, const VknStageWeights* w) {
    return w->ft_w && w->ft_wT && w->ft_b && w->dyn_w && w->dyn_b && w->fc_mask_w && w->fc_mask_b && d->n_mask_fcs > 0;
}

struct PrepItem {
    const float* src;
    int nout, k;
    const void** dst;
};

// enumerates (weight, shape, slot) in carve order; returns the number of items
int prep_items(const VknDims* d, const VknStageWeights* w, PrepW* p, PrepItem* it) {
    const int C = d->C, FF = d->ff;
    int n = 0;
    auto add = [&](const float* src, int nout, int k, const void** dst) {
        *dst = nullptr;
        if (src) it[n++] = PrepItem{src, nout, k, dst};
    };
    add(w->ft_w, C, C, &p->ft); add(w->ft_wT, C, C, &p->ftT);
    add(w->dyn_w, 2 * C, C, &p->dyn); add(w->inp_w, 2 * C, C, &p->inp);
    add(w->ig_w, C, C, &p->ig); add(w->ug_w, C, C, &p->ug); add(w->fc_w, C, C, &p->fc);
    add(w->attn_in_w, 3 * C, C, &p->attn_in); add(w->attn_out_w, C, C, &p->attn_out);
    add(w->ffn1_w, FF, C, &p->ffn1); add(w->ffn2_w, C, FF, &p->ffn2);
    for (int i = 0; i < VKN_MAX_FCS; ++i) add(i < d->n_cls_fcs ? w->cls_fc_w[i] : nullptr, C, C, &p->cls_fc[i]);
    add(w->fc_cls_w, d->ncls, C, &p->fc_cls);
    for (int i = 0; i < VKN_MAX_FCS; ++i) add(i < d->n_mask_fcs ? w->mask_fc_w[i] : nullptr, C, C, &p->mask_fc[i]);
    add(w->fc_mask_w, C, C, &p->fc_mask);
    // cross-attention: q rows and k/v rows of the packed in_proj are separate GEMMs -> separate tile images
    add(w->pa_in_w, C, C, &p->pa_in); add(w->pa_in_w ? w->pa_in_w + (size_t)C * C : nullptr, 2 * C, C, &p->pa_in_kv);
    add(w->pa_out_w, C, C, &p->pa_out);
    add(w->lffn1_w, FF, C, &p->lffn1); add(w->lffn2_w, C, FF, &p->lffn2);
    return n;
}

// assigns slots inside `base` (may be null: size query); returns total bytes
size_t carve_prepared(const VknDims* d, const VknStageWeights* w, char* base, PrepW* p, PrepItem* it, int* n_out) {
    const int n = prep_items(d, w, p, it);
    Carver c{base, 0};
    for (int i = 0; i < n; ++i) *it[i].dst = c.take<char>(vkn_split_w3_bytes(it[i].nout, it[i].k));
    p->dynft = p->dec = nullptr;
    p->dynft32 = p->bcnt = p->dec32 = p->decb = p->dvec = p->kb0 = p->fmT = p->chain_consts = nullptr;
    if (has_composites(d, w)) {
        const size_t C = d->C;
        p->dynft = c.take<char>(vkn_split_w3_bytes(2 * d->C, d->C));
        p->dec = c.take<char>(vkn_split_w3_bytes(d->C, d->C));
        p->dynft32 = c.take<float>(2 * C * C);
        p->bcnt = c.take<float>(2 * C);
        p->dec32 = c.take<float>(C * C);
        p->decb = c.take<float>(C);
        p->dvec = c.take<float>(C);
        p->kb0 = c.take<float>(4);
        p->fmT = c.take<float>(C * C);
        if (d->C == 256 && d->ff <= 2048) p->chain_consts = c.take<float>(vkn_chain_consts_floats());
    }
    for (int i = 0; i < VKN_H2_COUNT; ++i) p->h2[i] = nullptr;
    p->h2_scale = nullptr;
    p->h2_scratch = nullptr;
    // (shape conditions only: in a size query — base == NULL — every slot pointer is NULL)
    if (has_composites(d, w) && d->C == 256 && d->ff <= 2048 && d->ff % 256 == 0 && d->n_cls_fcs == 1 && d->n_mask_fcs == 1 && w->inp_w && w->ig_w && w->ug_w && w->fc_w &&
        w->attn_in_w && w->attn_out_w && w->ffn1_w && w->ffn2_w && w->cls_fc_w[0] && w->mask_fc_w[0]) {
        const int C = d->C, FF = d->ff;
        const int nout[VKN_H2_COUNT] = {2 * C, 2 * C, 2 * C, C, C, C, 3 * C, C, FF, C, C, C, w->fc_cls_w ? d->ncls : 0, C};
        const int kk[VKN_H2_COUNT] = {C, C, C, C, C, C, C, C, C, FF, C, C, C, C};
        for (int i = 0; i < VKN_H2_COUNT; ++i)
            if (nout[i] > 0) p->h2[i] = c.take<char>(vkn_split_h2_bytes(nout[i], kk[i]));
        p->h2_scale = c.take<float>(VKN_H2_COUNT * 8);
        p->h2_scratch = c.take<unsigned>(4);
    }
    if (n_out) *n_out = n;
    return (c.off + 255) & ~(size_t)255;
}

// hidden splits of the fused FFN: enough workgroups to fill the chip, each with whole 256-wide hidden chunks; 0 = not applicable
int ffn_hsplit(int M, int FF) {
    if (FF % 256 != 0) return 0;
    const int chunks = FF / 256, rt = (M + 31) / 32;
    int hs = 1;
    while (hs < chunks && rt * hs < 256 && chunks % (hs * 2) == 0) hs *= 2;
    return hs;
}

int ffn_ksplit(int M, int K) {
    const int rt = (M + 31) / 32, ktiles = K / 32;
    int ks = 256 / (rt > 0 ? rt : 1);
    if (ks > 16) ks = 16;
    if (ks > ktiles) ks = ktiles;
    if (ks < 1) ks = 1;
    return ks;
}

size_t carve_stage(const VknDims* d, char* base, StageWs* s) {
    Carver c{base, 0};
    const size_t B = d->B, N = d->N, C = d->C, P = (size_t)d->H * d->W, M = B * N, FF = d->ff;
    const size_t G = vkn_gather_groups(d->B, (int)P), NPT = npt_of(d->N);
    s->status = c.take<int>(64);   // offset 0 of every stage / head / chain workspace (256-byte header)
    s->part = c.take<float>(B * G * NPT * C);
    s->cntp = c.take<float>(B * G * NPT);
    s->xraw = c.take<float>(M * C);
    s->cnt = c.take<float>(M);
    s->xfeat = c.take<float>(M * C);
    s->params = c.take<float>(M * 2 * C);
    s->inputf = c.take<float>(M * 2 * C);
    s->ig = c.take<float>(M * 2 * C);  // fused path: [M][2C] = input gate | update gate
    s->ug = c.take<float>(M * C);
    s->f = c.take<float>(M * C);
    s->obj1 = c.take<float>(M * C);
    s->qkv = c.take<float>(M * 3 * C);
    s->ao = c.take<float>(M * C);
    s->obj2 = c.take<float>(M * C);
    s->h = c.take<float>(M * FF);
    {
        // split-K / hidden-split partials: for the whole batch, and for ONE frame (the frame-sequential last stage of the
        // previous_link heads runs the chain with B = 1 inside a B-frame call, with that shape's own split factors)
        const size_t ks = (size_t)ffn_ksplit((int)M, (int)FF), hs = (size_t)ffn_hsplit((int)M, (int)FF);
        const size_t ks1 = (size_t)ffn_ksplit((int)N, (int)FF), hs1 = (size_t)ffn_hsplit((int)N, (int)FF);
        const size_t k4 = 4;   // the few-row chain's z-split FFN (vkn_ksplit.hip) stores up to four partial results
        const size_t a = (ks > hs ? (ks > k4 ? ks : k4) : (hs > k4 ? hs : k4)) * M * C, b1 = (ks1 > hs1 ? (ks1 > k4 ? ks1 : k4) : (hs1 > k4 ? hs1 : k4)) * N * C;
        s->partial = c.take<float>(a > b1 ? a : b1);
    }
    s->t1 = c.take<float>(M * C);
    s->t2 = c.take<float>(M * C);
    s->maskfeat = c.take<float>(M * C);
    s->kb = c.take<float>(M);
    s->kern32 = c.take<float>(M * C);
    s->lq = c.take<float>(M * C);
    s->lkv = c.take<float>(M * 2 * C);
    s->lobj = c.take<float>(M * C);
    s->lupd = c.take<float>(M * C);
    s->lf = c.take<float>(M * C);
    s->kfh = c.take<_Float16>(B * NPT * C);
    s->kfl = c.take<_Float16>(B * NPT * C);
    return (c.off + 255) & ~(size_t)255;
}

int check_dims(const VknDims* d) {
    if (!d) return VKN_E_ARG;
    if (d->B <= 0 || d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->heads <= 0 || d->ff <= 0 || d->ncls <= 0)
        return VKN_E_ARG;
    if (d->C % 32 != 0 || d->C > 256) return VKN_E_SHAPE;
    if (d->C % d->heads != 0 || d->C / d->heads > 64 || (d->C % 8) != 0 || d->C / 8 > 64) return VKN_E_SHAPE;
    {
        const int hd = d->C / d->heads, hd8 = d->C / 8;  // head dims of `attention` and of the link attention (8 heads)
        if (hd < 4 || (hd & (hd - 1)) || hd8 < 4 || (hd8 & (hd8 - 1))) return VKN_E_SHAPE;
    }
    if (d->ff % 32 != 0) return VKN_E_SHAPE;
    if (d->N > 256) return VKN_E_SHAPE;
    if (d->ncls > 256) return VKN_E_SHAPE;
    if (d->n_cls_fcs < 0 || d->n_cls_fcs > VKN_MAX_FCS || d->n_mask_fcs < 0 || d->n_mask_fcs > VKN_MAX_FCS) return VKN_E_SHAPE;
    return VKN_OK;
}

// storage type of x selected by the flags (0 fp32, 1 fp16, 2 bf16)
inline int xdt_of(unsigned flags) { return (flags & VKN_FLAG_X_F16) ? 1 : ((flags & VKN_FLAG_X_BF16) ? 2 : 0); }

VknEpi mk_epi(const VknDims* d) {
    VknEpi e{};
    e.eps = d->ln_eps;
    return e;
}

#define VKN_TRY(expr)             \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != VKN_OK) return rc_; \
    } while (0)

// FFN: out = LN(in + W2 relu(W1 in + b1) + b2)        (mmcv FFN + following LayerNorm)
int run_ffn(const VknDims* d, const StageWs& s, const float* in, const float* w1, const void* w1s, const float* b1,
            const float* w2, const void* w2s, const float* b2, const float* nw, const float* nb, float* out, hipStream_t st) {
    const int M = d->B * d->N, C = d->C, FF = d->ff;
    VknEpi e = mk_epi(d);
    // both Linears in one kernel (hidden activations stay on chip) when the weights are pre-split and the shape allows it
    int hsplit = ffn_hsplit(M, FF);
    {
        const int hs_dbg = vkn_dbg_env("VKN_FFN_HS", 0);  // debug build: smaller hidden split (A/B)
        if (hs_dbg > 0 && hs_dbg <= hsplit) hsplit = hs_dbg;
    }
    if (w1s && w2s && C == 256 && hsplit > 0 && vkn_dbg_env("VKN_FFN_FUSED", 1) != 0) {
        e.bias = b2; e.resid = in; e.ldr = C; e.ln_w = nw; e.ln_b = nb; e.out = out; e.ldo = C;
        return vkn_launch_ffn_fused(in, C, w1s, b1, w2s, M, C, FF, hsplit, s.partial, e, st);
    }
    e.bias = b1; e.act = 1; e.out = s.h; e.ldo = FF;
    VKN_TRY(vkn_launch_gemm(in, nullptr, C, w1, w1s, M, C, FF, 1, nullptr, e, st));
    e = mk_epi(d);
    e.bias = b2; e.resid = in; e.ldr = C; e.ln_w = nw; e.ln_b = nb; e.out = out; e.ldo = C;
    return vkn_launch_gemm(s.h, nullptr, FF, w2, w2s, M, FF, C, ffn_ksplit(M, FF), s.partial, e, st);
}

// attention block: out = LN(identity + out_proj(softmax(q k^T / sqrt(hd)) v)); q from `qsrc`, k/v from `kvsrc`
int run_attention(const VknDims* d, const StageWs& s, const float* qsrc, const float* kvsrc, int heads, const float* in_w,
                  const void* in_ws, const void* in_kv_ws, const float* in_b, const float* out_w, const void* out_ws, const float* out_b,
                  const float* nw, const float* nb, float* out, hipStream_t st) {
    const int M = d->B * d->N, C = d->C, hd = C / heads;
    VknEpi e = mk_epi(d);
    if (qsrc == kvsrc) {  // self-attention: one packed in_proj GEMM
        e.bias = in_b; e.out = s.qkv; e.ldo = 3 * C;
        VKN_TRY(vkn_launch_gemm(qsrc, nullptr, C, in_w, in_ws, M, C, 3 * C, 1, nullptr, e, st));
        VKN_TRY(vkn_launch_attn(s.qkv, 3 * C, s.qkv + C, s.qkv + 2 * C, 3 * C, s.ao, C, d->B, d->N, d->N, heads, hd, st));
    } else {
        e.bias = in_b; e.out = s.lq; e.ldo = C;
        VKN_TRY(vkn_launch_gemm(qsrc, nullptr, C, in_w, in_ws, M, C, C, 1, nullptr, e, st));
        e = mk_epi(d);
        e.bias = in_b + C; e.out = s.lkv; e.ldo = 2 * C;
        VKN_TRY(vkn_launch_gemm(kvsrc, nullptr, C, in_w + (size_t)C * C, in_kv_ws, M, C, 2 * C, 1, nullptr, e, st));
        VKN_TRY(vkn_launch_attn(s.lq, C, s.lkv, s.lkv + C, 2 * C, s.ao, C, d->B, d->N, d->N, heads, hd, st));
    }
    e = mk_epi(d);
    e.bias = out_b; e.resid = qsrc; e.ldr = C; e.ln_w = nw; e.ln_b = nb; e.out = out; e.ldo = C;
    return vkn_launch_gemm(s.ao, nullptr, C, out_w, out_ws, M, C, C, 1, nullptr, e, st);
}

// KernelUpdator.forward                                        knet/kernel_updator.py:56-93
// `xraw`/`cnt` non-null (and composites prepared): dynamic_layer consumes the raw gather through W_dyn.W_ft (+ cnt (x) W_dyn.b_ft).
int run_updator(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* xfeat, const float* xraw,
                const float* cnt, const float* obj_in, float* out, const StageWs& s, hipStream_t st) {
    const int C = d->C, M = d->B * d->N;
    // with C == 256 param_out / input_out are exactly the second 256-column tile of their GEMM: their LayerNorms (:79-80) run
    // in that tile's epilogue, and the mix (:83-88) becomes the A prologue of fc_layer -> no standalone mix kernel.
    const bool fuse = (C == 256);
    // dynamic_layer(x_feat) and input_layer(kernels) are independent: one grouped launch                     (:59, :65-66)
    VknGemmProb pr[2];
    VknEpi e = mk_epi(d); e.out = s.params; e.ldo = 2 * C;
    if (fuse) { e.ln_w = w->norm_out_w; e.ln_b = w->norm_out_b; e.ln_from_col = C; }
    if (xraw && pw.dynft) {
        e.bias = pw.bcnt; e.rowscale = cnt; e.bias2 = w->dyn_b;
        pr[0] = VknGemmProb{xraw, nullptr, nullptr, nullptr, C, pw.dynft32, pw.dynft, 2 * C, e};
    } else {
        e.bias = w->dyn_b;
        pr[0] = VknGemmProb{xfeat, nullptr, nullptr, nullptr, C, w->dyn_w, pw.dyn, 2 * C, e};
    }
    e = mk_epi(d); e.bias = w->inp_b; e.out = s.inputf; e.ldo = 2 * C;
    if (fuse) { e.ln_w = w->inorm_out_w; e.ln_b = w->inorm_out_b; e.ln_from_col = C; }
    pr[1] = VknGemmProb{obj_in, nullptr, nullptr, nullptr, C, w->inp_w, pw.inp, 2 * C, e};
    VKN_TRY(vkn_launch_gemm_group(pr, 2, M, C, 1, nullptr, st));
    // gate = input_in * param_in (:70) as the GEMM's A prologue; both gates = sigmoid(LN(linear(gate))) in one launch (:74-78)
    float* ig = s.ig;
    float* ug = fuse ? s.ig + C : s.ug;
    const int ldg = fuse ? 2 * C : C;
    e = mk_epi(d); e.bias = w->ig_b; e.ln_w = w->inorm_in_w; e.ln_b = w->inorm_in_b; e.act = 2; e.out = ig; e.ldo = ldg;
    pr[0] = VknGemmProb{s.inputf, s.params, nullptr, nullptr, 2 * C, w->ig_w, pw.ig, C, e};
    e = mk_epi(d); e.bias = w->ug_b; e.ln_w = w->norm_in_w; e.ln_b = w->norm_in_b; e.act = 2; e.out = ug; e.ldo = ldg;
    pr[1] = VknGemmProb{s.inputf, s.params, nullptr, nullptr, 2 * C, w->ug_w, pw.ug, C, e};
    VKN_TRY(vkn_launch_gemm_group(pr, 2, M, C, 1, nullptr, st));
    e = mk_epi(d); e.bias = w->fc_b; e.ln_w = w->fc_norm_w; e.ln_b = w->fc_norm_b; e.act = 1; e.out = out; e.ldo = C;
    if (fuse) {
        // features = update_gate * norm_out(param_out) + input_gate * input_norm_out(input_out)            (:83-88)
        pr[0] = VknGemmProb{ug, s.params + C, ig, s.inputf + C, 2 * C, w->fc_w, pw.fc, C, e};
        return vkn_launch_gemm_group(pr, 1, M, C, 1, nullptr, st);                                  // :90-92
    }
    VKN_TRY(vkn_launch_ku_mix(s.params, s.inputf, ig, ug, w->norm_out_w, w->norm_out_b, w->inorm_out_w, w->inorm_out_b,
                              d->ln_eps, s.f, M, C, st));                                          // :79-88
    return vkn_launch_gemm(s.f, nullptr, C, w->fc_w, pw.fc, M, C, C, 1, nullptr, e, st);           // :90-92
}

// link block (see StageOpts): video tracking link, previous_type == "ffn"      knet/video/kernel_update_head.py:394-415
//                                  previous_type == "update" / "update_obj"      :417-476   (updator on x_feat / on obj_feat)
//                                  previous_link == "update_dynamic_cov"         :324-348   (updator on x_feat; out replaces obj_in)
//                                  previous_link == "link_atten"                 :350-372
bool link_ks_ok(const VknDims* d, const VknStageWeights* w, const PrepW& pw, unsigned flags);
int run_link_ks(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* cur, const float* kv, float* out,
                const StageWs& s, hipStream_t st);
int run_link(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* cur, const float* prev,
             float* out, const StageWs& s, hipStream_t st, const float* update_feature = nullptr, unsigned flags = 0) {
    if (!w->pa_in_w || !w->lffn1_w) return VKN_E_ARG;
    const float* kv = prev;
    if (update_feature) {   // (the stage's OWN weights also carry an updator — the main one: only an update feature selects it)
        if (!w->dyn_w) return VKN_E_ARG;
        StageWs su = s;
        su.f = s.lf;  // (C != 256: the unfused mix buffer; `f` itself is a cls / mask branch scratch of the main stream)
        VKN_TRY(run_updator(d, w, pw, update_feature, nullptr, nullptr, prev, s.lupd, su, st));
        kv = s.lupd;
    }
    if (link_ks_ok(d, w, pw, flags)) return run_link_ks(d, w, pw, cur, kv, out, s, st);   // few rows: column-spread phases (vkn_ksplit.hip)
    VKN_TRY(run_attention(d, s, cur, kv, 8, w->pa_in_w, pw.pa_in, pw.pa_in_kv, w->pa_in_b, w->pa_out_w, pw.pa_out, w->pa_out_b,
                          w->pa_norm_w, w->pa_norm_b, s.t1, st));                                             // _num_head = 8 (:165)
    return run_ffn(d, s, s.t1, w->lffn1_w, pw.lffn1, w->lffn1_b, w->lffn2_w, pw.lffn2, w->lffn2_b, w->lffn_norm_w,
                   w->lffn_norm_b, out, st);
}

int carve_pw(const VknDims* d, const VknStageWeights* w, unsigned flags, PrepW* pw) {
    *pw = PrepW{};
    if (w->prepared && !(flags & VKN_FLAG_EXACT_GEMM)) {
        PrepItem items[40];
        if (carve_prepared(d, w, static_cast<char*>(const_cast<void*>(w->prepared)), pw, items, nullptr) > w->prepared_bytes)
            return VKN_E_WORKSPACE;
    }
    return VKN_OK;
}

// The LAST stage's logits decode from the planes in `s` (+ the caller's xS up-scaled output when up_out is given): in chunks of
// up_chunk frames, each chunk's upsample right behind its decode, so the upsample reads logits that are still in the memory-side
// cache.  Odd H*W / VKN_FLAG_REF_KERNELS: the exact-fp32 kernel on s.kern32.
int final_decode(const VknDims* d, const float* x, const StageWs& s, const float* kb, float* masks_out, unsigned flags,
                 hipStream_t st, hipEvent_t prof0, hipEvent_t prof1, float* up_out, int up_stride, int up_chunk, bool* up_done) {
    const int B = d->B, N = d->N, C = d->C, P = d->H * d->W;
    if ((flags & VKN_FLAG_REF_KERNELS) || (P & 1)) return vkn_launch_decode_ref(x, s.kern32, kb, masks_out, B, N, C, P, st);
    const int ch = (up_out && up_chunk > 0 && up_chunk < B) ? up_chunk : B;
    const size_t NPTC = (size_t)npt_of(N) * C;
    for (int b0 = 0; b0 < B; b0 += ch) {
        const int bn = (B - b0 < ch) ? B - b0 : ch;
        const float* xb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + (size_t)b0 * C * P * (xdt_of(flags) ? 2 : 4));
        if (b0 == 0 && prof0 && hipEventRecord(prof0, st) != hipSuccess) return VKN_E_LAUNCH;  // the (first) decode launch alone
        VKN_TRY(vkn_launch_decode(xb, s.kfh + b0 * NPTC, s.kfl + b0 * NPTC, kb ? kb + (size_t)b0 * N : nullptr,
                                  masks_out + (size_t)b0 * N * P, bn, N, C, P, st, xdt_of(flags)));
        if (b0 == 0 && prof1 && hipEventRecord(prof1, st) != hipSuccess) return VKN_E_LAUNCH;
        if (ch < B)
            VKN_TRY(vkn_launch_upsample(masks_out + (size_t)b0 * N * P,
                                        (flags & VKN_FLAG_SCALED_F16) ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(up_out) + (size_t)b0 * N * P * up_stride * up_stride)
                                                                      : up_out + (size_t)b0 * N * P * up_stride * up_stride,
                                        bn * N, d->H, d->W, up_stride, st, (flags & VKN_FLAG_SCALED_F16) ? 1 : 0));
    }
    if (ch < B && up_done) *up_done = true;
    return VKN_OK;
}

// The persistent row-owner chain (vkn_chain.hip) covers the shipped shape: C == 256, one cls / mask FC, an FFN whose width is a
// multiple of 256 (<= 2048), composite (feat_transform-folded) pre-split weights.  Everything else — and VKN_FLAG_CHAIN_LAUNCHES
// (A/B) — takes the launch-per-GEMM path below.
inline int persistent_min_row_tiles(const PrepW& pw, unsigned flags) {
    return (!(flags & VKN_FLAG_CHAIN_BF16X3) && pw.h2_scale && pw.h2[VKN_H2_DYNFT]) ? 22 : 64;
}
inline unsigned pw_off(const VknStageWeights* w, const void* p) {
    return (unsigned)(static_cast<const char*>(p) - static_cast<const char*>(w->prepared));
}
bool chain_fast_ok(const VknDims* d, const VknStageWeights* w, const PrepW& pw, unsigned flags, bool have_cls) {
    if (flags & (VKN_FLAG_CHAIN_LAUNCHES | VKN_FLAG_EXACT_GEMM)) return false;
    if (vkn_dbg_env("VKN_CHAIN_LAUNCHES", 0)) return false;
    // Policy (profiles/r04_chain_ab.txt): a row-owner workgroup streams ALL of a stage's weights (12 MB) through its CU, ~160 us per
    // stage however many rows there are, while the launch-per-GEMM chain spreads the tile stream over the chip and costs
    // 124 / 149 / 202 / 341 us at 117 / 1872 / 3744 / 7488 rows: the persistent kernels win from ~64 row tiles (2048 rows) on.
    // Round 5: on the two-term fp16 split the row owners stream 8 MB instead of 12 — 112 us per stage at any row count against
    // 108 / 124 / 127 / 160 / 175 us of the launch-per-GEMM chain at 936 / 1404 / 1872 / 2340 / 3744 rows (profiles/r05_chain_forms.txt):
    // chain alone they win from ~40 row tiles on, but inside a head step earlier — two launches instead of ten and the weight warm-up of the
    // preceding gather reduction: whole steps at 5 / 6 / 7 / 8 / 9 / 10 frames per call 0.857 / 0.947 / 1.057 / 1.156 / 1.192 / 1.285 ms against
    // 0.845 / 0.953 / 1.061 / 1.187 / 1.262 / 1.370 on the launch-per-GEMM chain (profiles/r05_chain_forms.txt) => from 22 row tiles (6 frames
    // of 117 kernels) on; on the bf16 split (VKN_FLAG_CHAIN_BF16X3) from 64 as before.
    if (!(flags & VKN_FLAG_CHAIN_PERSISTENT) && !vkn_dbg_env("VKN_CHAIN_PERSISTENT", 0) &&
        (d->B * d->N + 31) / 32 < persistent_min_row_tiles(pw, flags))
        return false;
    if (d->C != 256 || d->n_cls_fcs != 1 || d->n_mask_fcs != 1 || d->ff % 256 != 0 || d->ff > 2048) return false;
    if (!w->prepared || w->prepared_bytes >= (1ull << 31)) return false;
    if (!pw.chain_consts || !pw.dynft || !pw.dyn || !pw.dec || !pw.inp || !pw.ig || !pw.ug || !pw.fc || !pw.attn_in || !pw.attn_out || !pw.ffn1 ||
        !pw.ffn2 || !pw.cls_fc[0] || !pw.mask_fc[0])
        return false;
    if (have_cls && !pw.fc_cls) return false;
    if (!w->ffn1_w || !w->cls_ln_w[0] || !w->mask_ln_w[0]) return false;
    return true;
}
// the persistent chain runs on the two-term fp16 split (vkn_chain_h2.hip) wherever its images were prepared; VKN_FLAG_CHAIN_BF16X3 opts out
inline bool chain_h2(const PrepW& pw, unsigned flags, bool have_cls) {
    return !(flags & VKN_FLAG_CHAIN_BF16X3) && pw.h2_scale && pw.h2[VKN_H2_DYNFT] && pw.h2[VKN_H2_DEC] && (!have_cls || pw.h2[VKN_H2_FCCLS]);
}

// (ii) + the FC branches as three launches: k_chain_a, the attention, k_chain_c.  `a0` / `rowscale`: the raw gather + pixel counts
// (composite dynamic weights) or x_feat (rowscale NULL).  The decode kernels leave as f16 planes (s.kfh / s.kfl) or as fp32 (kern32_out).
int run_chain_fast(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* a0, bool a0_raw, const float* cnt,
                   const float* obj_in, float* obj_out, float* cls_logits, bool cls_sigmoid, float* kern32_out, const StageWs& s,
                   hipStream_t st, bool h2 = false) {
    const int M = d->B * d->N, C = d->C;
    VknChainA a{};
    a.a0 = a0; a.obj_in = obj_in; a.rowscale = a0_raw ? cnt : nullptr;
    a.wbase = w->prepared; a.wbytes = w->prepared_bytes;
    if (h2) {
        a.off_dyn = pw_off(w, pw.h2[a0_raw ? VKN_H2_DYNFT : VKN_H2_DYN]);
        a.off_inp = pw_off(w, pw.h2[VKN_H2_INP]); a.off_ig = pw_off(w, pw.h2[VKN_H2_IG]); a.off_ug = pw_off(w, pw.h2[VKN_H2_UG]);
        a.off_fc = pw_off(w, pw.h2[VKN_H2_FC]); a.off_in = pw_off(w, pw.h2[VKN_H2_IN]);
    } else {
        a.off_dyn = pw_off(w, a0_raw ? pw.dynft : pw.dyn);
        a.off_inp = pw_off(w, pw.inp); a.off_ig = pw_off(w, pw.ig); a.off_ug = pw_off(w, pw.ug); a.off_fc = pw_off(w, pw.fc);
        a.off_in = pw_off(w, pw.attn_in);
    }
    a.consts = pw.chain_consts;
    a.eps = d->ln_eps; a.M = M; a.obj1 = s.obj1; a.qkv = s.qkv; a.status = s.status;
    VKN_TRY(h2 ? vkn_launch_chain_a_h2(a, st) : vkn_launch_chain_a(a, st));
    VKN_TRY(vkn_launch_attn(s.qkv, 3 * C, s.qkv + C, s.qkv + 2 * C, 3 * C, s.ao, C, d->B, d->N, d->N, d->heads, C / d->heads, st));
    VknChainC c{};
    c.ao = s.ao; c.obj1 = s.obj1; c.wbase = w->prepared; c.wbytes = w->prepared_bytes;
    if (h2) {
        c.off_out = pw_off(w, pw.h2[VKN_H2_OUT]); c.off_ffn1 = pw_off(w, pw.h2[VKN_H2_FFN1]); c.off_ffn2 = pw_off(w, pw.h2[VKN_H2_FFN2]);
        c.off_clsfc = pw_off(w, pw.h2[VKN_H2_CLSFC]); c.off_maskfc = pw_off(w, pw.h2[VKN_H2_MASKFC]);
        c.off_fccls = pw.h2[VKN_H2_FCCLS] ? pw_off(w, pw.h2[VKN_H2_FCCLS]) : 0u; c.off_dec = pw_off(w, pw.h2[VKN_H2_DEC]);
    } else {
        c.off_out = pw_off(w, pw.attn_out); c.off_ffn1 = pw_off(w, pw.ffn1); c.off_ffn2 = pw_off(w, pw.ffn2);
        c.off_clsfc = pw_off(w, pw.cls_fc[0]); c.off_maskfc = pw_off(w, pw.mask_fc[0]);
        c.off_fccls = pw.fc_cls ? pw_off(w, pw.fc_cls) : 0u; c.off_dec = pw_off(w, pw.dec);
    }
    c.consts = pw.chain_consts; c.kb0 = pw.kb0;
    c.ff = d->ff; c.ncls = d->ncls; c.cls_sigmoid = cls_sigmoid ? 1 : 0; c.eps = d->ln_eps; c.M = M;
    c.obj_out = obj_out; c.cls_out = (w->fc_cls_w && cls_logits) ? cls_logits : nullptr; c.kb_out = s.kb;
    if (kern32_out) c.kern_out = kern32_out;
    else { c.plane_hi = s.kfh; c.plane_lo = s.kfl; }
    c.rows_per_frame = d->N; c.NPT = npt_of(d->N); c.status = s.status;
    return h2 ? vkn_launch_chain_c_h2(c, st) : vkn_launch_chain_c(c, st);
}

// Row-count policy of the three chain forms (profiles/r05_chain_forms.txt; chain alone, us per stage at 117 / 234 / 351 / 468 / 585 / 936 /
// 1872 / 3744 rows): few-row 77 / 78 / 87 / 87 / 109 / 117 / 213 / 368, launch-per-GEMM 105 / 108 / 108 / 107 / 107 / 108 / 128 / 174,
// persistent 158 at any row count -> few-row up to 16 row tiles (4 frames of 117 kernels), launch-per-GEMM up to 63, persistent from 64 on.
#define VKN_KS_MAX_ROW_TILES 16
// The few-row chain (vkn_ksplit.hip): same shape conditions as the persistent chain, at most VKN_KS_MAX_ROW_TILES row tiles (or VKN_FLAG_CHAIN_KSPLIT),
// every vector it reads with 16-byte loads aligned (parameters that are views into a packed buffer may not be).
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool chain_ks_ok(const VknDims* d, const VknStageWeights* w, const PrepW& pw, unsigned flags, bool have_cls) {
    if (flags & (VKN_FLAG_CHAIN_LAUNCHES | VKN_FLAG_CHAIN_PERSISTENT | VKN_FLAG_EXACT_GEMM)) return false;
    if (vkn_dbg_env("VKN_CHAIN_LAUNCHES", 0) || vkn_dbg_env("VKN_CHAIN_PERSISTENT", 0) || !vkn_dbg_env("VKN_CHAIN_KSPLIT", 1)) return false;
    if (!(flags & VKN_FLAG_CHAIN_KSPLIT) && (d->B * d->N + 31) / 32 > VKN_KS_MAX_ROW_TILES) return false;
    if (d->C != 256 || d->n_cls_fcs != 1 || d->n_mask_fcs != 1 || d->ff % 256 != 0 || d->ff > 2048) return false;
    if (d->ff % 512 != 0 && d->ff > 1024) return false;   // at most four z-chunks of the FFN's second Linear
    if (!w->prepared || w->prepared_bytes >= (1ull << 31)) return false;
    if (!pw.dynft || !pw.dyn || !pw.dec || !pw.inp || !pw.ig || !pw.ug || !pw.fc || !pw.attn_in || !pw.attn_out || !pw.ffn1 || !pw.ffn2 ||
        !pw.cls_fc[0] || !pw.mask_fc[0])
        return false;
    if (have_cls && !pw.fc_cls) return false;
    if (!w->ffn1_w || !w->cls_ln_w[0] || !w->mask_ln_w[0]) return false;
    const void* v16[] = {w->inorm_in_w, w->inorm_in_b, w->norm_in_w, w->norm_in_b, w->norm_out_w, w->norm_out_b, w->inorm_out_w, w->inorm_out_b,
                         w->fc_norm_w, w->fc_norm_b, w->attn_norm_w, w->attn_norm_b, w->ffn_norm_w, w->ffn_norm_b, w->ffn2_b,
                         w->cls_ln_w[0], w->cls_ln_b[0], w->mask_ln_w[0], w->mask_ln_b[0], pw.dvec};
    for (const void* p : v16)
        if (!p || !al16(p)) return false;
    return true;
}

VknKsProb ks_prob(const float* a, int lda, const void* W, int Nout, int KT, float eps) {
    VknKsProb p{};
    p.pro.a[0] = a; p.pro.lda[0] = lda; p.pro.nsum = 1; p.pro.eps = eps;
    p.Wsplit = W; p.Nout = Nout; p.KT = KT;
    return p;
}

// (ii) + the FC branches as nine GEMM phases + the attention (reference lines: see run_updator / run_attention / run_ffn and the cls /
// mask branches of run_stage below).  A phase stores its RAW result; LayerNorm / ReLU / sigmoid / the gate and mix algebra run in the
// prologue of the phase that consumes it.  `obj_ready` (or NULL) is recorded where obj_out is final — before the last phase.
int run_chain_ks(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* a0, bool a0_raw, const float* cnt,
                 const float* obj_in, float* obj_out, float* cls_logits, bool cls_sigmoid, float* kern32_out, const StageWs& s,
                 hipStream_t st, hipEvent_t obj_ready) {
    const int M = d->B * d->N, C = d->C, FF = d->ff;
    const float eps = d->ln_eps;
    VknKsProb pr[2];
    // dynamic_layer(update feature) | input_layer(kernels)                                            knet/kernel_updator.py:59-66
    pr[0] = ks_prob(a0, C, a0_raw ? pw.dynft : pw.dyn, 2 * C, 8, eps);
    if (a0_raw) { pr[0].epi.bias = pw.bcnt; pr[0].epi.rowscale = cnt; pr[0].epi.bias2 = w->dyn_b; }
    else pr[0].epi.bias = w->dyn_b;
    pr[0].epi.out = s.params; pr[0].epi.ldo = 2 * C;
    pr[1] = ks_prob(obj_in, C, pw.inp, 2 * C, 8, eps);
    pr[1].epi.bias = w->inp_b; pr[1].epi.out = s.inputf; pr[1].epi.ldo = 2 * C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 2, 0, 0, 1, 0, M, st));
    // gate = input_in * parameters_in; input_gate | update_gate (raw)                                 :70-76
    pr[0] = ks_prob(s.inputf, 2 * C, pw.ig, C, 8, eps);
    pr[0].pro.a[1] = s.params; pr[0].pro.lda[1] = 2 * C;
    pr[1] = pr[0];
    pr[0].epi.bias = w->ig_b; pr[0].epi.out = s.ig; pr[0].epi.ldo = 2 * C;
    pr[1].Wsplit = pw.ug; pr[1].epi.bias = w->ug_b; pr[1].epi.out = s.ig + C; pr[1].epi.ldo = 2 * C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 2, 1, 0, 1, 0, M, st));
    // features = sigmoid(norm_in(update_gate)) * norm_out(param_out) + sigmoid(input_norm_in(input_gate)) * input_norm_out(input_out);
    // fc_layer (raw)                                                                                   :74-90
    pr[0] = ks_prob(s.ig, 2 * C, pw.fc, C, 8, eps);
    pr[0].pro.a[1] = s.ig + C; pr[0].pro.lda[1] = 2 * C;
    pr[0].pro.a[2] = s.params + C; pr[0].pro.lda[2] = 2 * C;
    pr[0].pro.a[3] = s.inputf + C; pr[0].pro.lda[3] = 2 * C;
    pr[0].pro.ln_w[0] = w->inorm_in_w; pr[0].pro.ln_b[0] = w->inorm_in_b;
    pr[0].pro.ln_w[1] = w->norm_in_w; pr[0].pro.ln_b[1] = w->norm_in_b;
    pr[0].pro.ln_w[2] = w->norm_out_w; pr[0].pro.ln_b[2] = w->norm_out_b;
    pr[0].pro.ln_w[3] = w->inorm_out_w; pr[0].pro.ln_b[3] = w->inorm_out_b;
    pr[0].epi.bias = w->fc_b; pr[0].epi.out = s.f; pr[0].epi.ldo = C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 2, 0, 1, 0, M, st));
    // obj1 = relu(fc_norm(.)) (side output); q | k | v = in_proj(obj1)                                :91-92, knet/det/kernel_update_head.py:206
    pr[0] = ks_prob(s.f, C, pw.attn_in, 3 * C, 8, eps);
    pr[0].pro.ln_w[0] = w->fc_norm_w; pr[0].pro.ln_b[0] = w->fc_norm_b; pr[0].pro.act = 1;
    pr[0].pro.side_out = s.obj1; pr[0].pro.ld_side = C;
    pr[0].epi.bias = w->attn_in_b; pr[0].epi.out = s.qkv; pr[0].epi.ldo = 3 * C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, 0, 1, 0, M, st));
    VKN_TRY(vkn_launch_attn(s.qkv, 3 * C, s.qkv + C, s.qkv + 2 * C, 3 * C, s.ao, C, d->B, d->N, d->N, d->heads, C / d->heads, st));
    // out_proj + identity (raw)                                                                        knet/det/kernel_update_head.py:206
    pr[0] = ks_prob(s.ao, C, pw.attn_out, C, 8, eps);
    pr[0].epi.bias = w->attn_out_b; pr[0].epi.resid = s.obj1; pr[0].epi.ldr = C; pr[0].epi.out = s.obj2; pr[0].epi.ldo = C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, 0, 1, 0, M, st));
    // obj2 = attention_norm(.) (side output, the FFN's residual); hidden = relu(W1 obj2 + b1)          :208, :214
    pr[0] = ks_prob(s.obj2, C, pw.ffn1, FF, 8, eps);
    pr[0].pro.ln_w[0] = w->attn_norm_w; pr[0].pro.ln_b[0] = w->attn_norm_b;
    pr[0].pro.side_out = s.t1; pr[0].pro.ld_side = C;
    pr[0].epi.bias = w->ffn1_b; pr[0].epi.act = 1; pr[0].epi.out = s.h; pr[0].epi.ldo = FF;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, 0, 1, 0, M, st));
    // W2 hidden, the contraction in z-chunks (partial results)                                         :214
    const int kpw = (FF % 512 == 0) ? 2 : 1, zch = FF / (256 * kpw);
    pr[0] = ks_prob(s.h, FF, pw.ffn2, C, FF / 32, eps);
    pr[0].epi.out = s.partial; pr[0].epi.ldo = C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, zch, kpw, (long long)M * C, M, st));
    // obj_out = ffn_norm(obj2 + sum partial + b2) (side output: the stage's kernels); cls_fcs[0] | mask_fcs[0] (raw, no bias)   :215-226
    pr[0] = ks_prob(s.partial, C, pw.cls_fc[0], C, 8, eps);
    pr[0].pro.nsum = zch; pr[0].pro.sum_stride = (long long)M * C;
    pr[0].pro.pbias = w->ffn2_b; pr[0].pro.presid = s.t1; pr[0].pro.ldr = C;
    pr[0].pro.ln_w[0] = w->ffn_norm_w; pr[0].pro.ln_b[0] = w->ffn_norm_b;
    pr[1] = pr[0];
    pr[0].pro.side_out = obj_out; pr[0].pro.ld_side = C;
    pr[0].epi.out = s.lkv; pr[0].epi.ldo = 2 * C;
    pr[1].Wsplit = pw.mask_fc[0]; pr[1].epi.out = s.lkv + C; pr[1].epi.ldo = 2 * C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 2, 0, 0, 1, 0, M, st));
    if (obj_ready && hipEventRecord(obj_ready, st) != hipSuccess) return VKN_E_LAUNCH;
    // fc_cls(relu(LN(cls branch))) (+ sigmoid on the last stage) | folded decode kernels + bias from relu(LN(mask branch))     :217-227, :247
    int np = 0;
    if (w->fc_cls_w && cls_logits) {
        pr[np] = ks_prob(s.lkv, 2 * C, pw.fc_cls, d->ncls, 8, eps);
        pr[np].pro.ln_w[0] = w->cls_ln_w[0]; pr[np].pro.ln_b[0] = w->cls_ln_b[0]; pr[np].pro.act = 1;
        pr[np].epi.bias = w->fc_cls_b; pr[np].epi.act = cls_sigmoid ? 2 : 0; pr[np].epi.out = cls_logits; pr[np].epi.ldo = d->ncls;
        ++np;
    }
    pr[np] = ks_prob(s.lkv + C, 2 * C, pw.dec, C, 8, eps);
    pr[np].pro.ln_w[0] = w->mask_ln_w[0]; pr[np].pro.ln_b[0] = w->mask_ln_b[0]; pr[np].pro.act = 1;
    pr[np].pro.dot_vec = pw.dvec; pr[np].pro.dot_bias = pw.kb0; pr[np].pro.dot_out = s.kb;
    pr[np].epi.bias = pw.decb; pr[np].epi.ldo = C;
    if (kern32_out) pr[np].epi.out = kern32_out;
    else { pr[np].epi.plane_hi = s.kfh; pr[np].epi.plane_lo = s.kfl; pr[np].epi.rows_per_frame = d->N; pr[np].epi.NPT = npt_of(d->N); }
    ++np;
    return vkn_launch_gemm_ks(pr, np, 0, 0, 1, 0, M, st);
}

// The link block LN(FFN(LN(cur + MHA_8(cur, kv)))) on the few-row kernels: q | k,v projections (one grouped launch), the attention,
// out_proj + identity (raw), FFN first Linear with attention_previous_norm in its prologue (side output: the FFN's residual), the
// second Linear z-split, and the closing LayerNorm as a row epilogue over the partial sums.   knet/video/kernel_update_head.py:394-415
bool link_ks_ok(const VknDims* d, const VknStageWeights* w, const PrepW& pw, unsigned flags) {
    if (flags & (VKN_FLAG_CHAIN_LAUNCHES | VKN_FLAG_CHAIN_PERSISTENT | VKN_FLAG_EXACT_GEMM)) return false;
    if (vkn_dbg_env("VKN_CHAIN_LAUNCHES", 0) || !vkn_dbg_env("VKN_CHAIN_KSPLIT", 1)) return false;
    if (!(flags & VKN_FLAG_CHAIN_KSPLIT) && (d->B * d->N + 31) / 32 > VKN_KS_MAX_ROW_TILES) return false;
    if (d->C != 256 || d->ff % 256 != 0 || d->ff > 2048 || (d->ff % 512 != 0 && d->ff > 1024)) return false;
    if (!pw.pa_in || !pw.pa_in_kv || !pw.pa_out || !pw.lffn1 || !pw.lffn2) return false;
    if (!w->pa_norm_w || !w->pa_norm_b || !al16(w->pa_norm_w) || !al16(w->pa_norm_b)) return false;
    return true;
}

int run_link_ks(const VknDims* d, const VknStageWeights* w, const PrepW& pw, const float* cur, const float* kv, float* out,
                const StageWs& s, hipStream_t st) {
    const int M = d->B * d->N, C = d->C, FF = d->ff;
    const float eps = d->ln_eps;
    VknKsProb pr[2];
    pr[0] = ks_prob(cur, C, pw.pa_in, C, 8, eps);                          // q = in_proj[:C](cur)
    pr[0].epi.bias = w->pa_in_b; pr[0].epi.out = s.lq; pr[0].epi.ldo = C;
    pr[1] = ks_prob(kv, C, pw.pa_in_kv, 2 * C, 8, eps);                    // k | v = in_proj[C:](kv)
    pr[1].epi.bias = w->pa_in_b ? w->pa_in_b + C : nullptr; pr[1].epi.out = s.lkv; pr[1].epi.ldo = 2 * C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 2, 0, 0, 1, 0, M, st));
    VKN_TRY(vkn_launch_attn(s.lq, C, s.lkv, s.lkv + C, 2 * C, s.ao, C, d->B, d->N, d->N, 8, C / 8, st));        // _num_head = 8 (:165)
    pr[0] = ks_prob(s.ao, C, pw.pa_out, C, 8, eps);                        // out_proj + identity (raw)
    pr[0].epi.bias = w->pa_out_b; pr[0].epi.resid = cur; pr[0].epi.ldr = C; pr[0].epi.out = s.t1; pr[0].epi.ldo = C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, 0, 1, 0, M, st));
    pr[0] = ks_prob(s.t1, C, pw.lffn1, FF, 8, eps);                        // t = attention_previous_norm(.) (side output); relu(W1 t + b1)
    pr[0].pro.ln_w[0] = w->pa_norm_w; pr[0].pro.ln_b[0] = w->pa_norm_b;
    pr[0].pro.side_out = s.lf; pr[0].pro.ld_side = C;
    pr[0].epi.bias = w->lffn1_b; pr[0].epi.act = 1; pr[0].epi.out = s.h; pr[0].epi.ldo = FF;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, 0, 1, 0, M, st));
    const int kpw = (FF % 512 == 0) ? 2 : 1, zch = FF / (256 * kpw);
    pr[0] = ks_prob(s.h, FF, pw.lffn2, C, FF / 32, eps);                   // W2 hidden, z-split
    pr[0].epi.out = s.partial; pr[0].epi.ldo = C;
    VKN_TRY(vkn_launch_gemm_ks(pr, 1, 0, zch, kpw, (long long)M * C, M, st));
    VknEpi e = mk_epi(d);                                                   // link_ffn_norm(t + sum partial + b2)
    e.bias = w->lffn2_b; e.resid = s.lf; e.ldr = C; e.ln_w = w->lffn_norm_w; e.ln_b = w->lffn_norm_b; e.out = out; e.ldo = C;
    return vkn_launch_rowepi(s.partial, zch, M, C, e, st);
}

int run_stage(const VknDims* d, const VknStageWeights* w, const float* x, const float* obj_in, const float* masks_in,
              const float* prev_obj, float* cls_logits, float* masks_out, float* obj_out, float* x_feat_out,
              float* track_out, const StageWs& s, unsigned flags, hipStream_t st, const unsigned* bits_in = nullptr,
              unsigned* bits_out = nullptr, bool cls_sigmoid = false, bool gathered_in = false, bool gather_out = false,
              hipEvent_t prof0 = nullptr, hipEvent_t prof1 = nullptr, const float* xfeat_in = nullptr, float* kern_out = nullptr,
              float* kb_out = nullptr, hipEvent_t obj_ready = nullptr, float* up_out = nullptr, int up_stride = 0, int up_chunk = 0,
              bool* up_done = nullptr, const StageOpts* so = nullptr) {
    // xfeat_in / kern_out / kb_out (vkn_stage_chain_f32): the [B*N, C] chain alone — the caller supplies x_feat (already
    // feat-transformed and, for the clip-level VIS heads, merged over the frames of a clip) and receives the folded fp32 decode
    // kernels + bias instead of decoded masks; no gather and no decode are launched, x / masks_in / masks_out are unused.
    // bits_in / bits_out (fused head only): the stage hand-off as bit words instead of fp32 logits — the gather consumes
    // nothing but bit(logit >= thr), so intermediate stages never write the 15.3 MB / frame of logits.
    // gather_out / gathered_in (fused head, default): the hand-off is the NEXT stage's gather itself — this stage's decode and the
    // next stage's gather run as one pass over x (vkn_fused.hip) that leaves xraw / cnt in the shared workspace, where the next
    // stage (gathered_in) finds them; neither logits nor bit words exist.
    const int B = d->B, N = d->N, C = d->C, P = d->H * d->W, M = B * N;
    const bool ref = (flags & VKN_FLAG_REF_KERNELS) != 0;
    const bool chain_only = kern_out != nullptr;
    const bool ref_decode = ref || (P & 1) || chain_only;  // odd H*W: mask rows are not 8-byte aligned -> exact-fp32 FMA decode kernel
                                                           // (chain_only: fp32 folded kernels are the output)
    const bool has_ft = w->ft_w != nullptr;
    const int xdt = xdt_of(flags);
    const bool skip_decode = so && so->skip_decode;
    const bool pre_link = so && so->link_pre && so->prev_pre;
    const bool need_xfeat = (pre_link && so->link_pre->dyn_w) || (so && so->link_track && so->track_src == 1) || (so && so->keep_xfeat);  // (the link may run after the stage: side stream)
    auto decode_final = [&](const float* kb) -> int {
        return final_decode(d, x, s, kb, masks_out, flags, st, prof0, prof1, up_out, up_stride, up_chunk, up_done);
    };
    // half-storage x: the MFMA kernels only (whole 64-px tiles); the exact-fp32 reference kernels read fp32
    if (xdt && !chain_only && (ref || (P % 64) != 0)) return VKN_E_SHAPE;
    PrepW pw{};
    if (w->prepared && !(flags & VKN_FLAG_EXACT_GEMM)) {
        PrepItem items[40];
        if (carve_prepared(d, w, static_cast<char*>(const_cast<void*>(w->prepared)), &pw, items, nullptr) > w->prepared_bytes)
            return VKN_E_WORKSPACE;
    }

    // the gather's reduction warms the memory-side cache with the weights of the persistent chain that follows it
    const bool will_fast = !xfeat_in && pw.dynft && chain_fast_ok(d, w, pw, flags, w->fc_cls_w && cls_logits) && vkn_dbg_env("VKN_CHAIN_TOUCH", 1);
    const void* touch_own = will_fast ? w->prepared : nullptr;
    const size_t touch_own_bytes = will_fast ? w->prepared_bytes : 0;
    const void* touch_nx = (so && vkn_dbg_env("VKN_CHAIN_TOUCH", 1)) ? so->touch_next : nullptr;
    const size_t touch_nx_bytes = touch_nx ? so->touch_next_bytes : 0;
    // (i) mask gather                                        knet/det/kernel_update_head.py:190-195
    if (gathered_in || xfeat_in) {
        // s.xraw / s.cnt were produced by the previous stage's fused decode -> gather pass (or x_feat is given)
    } else if (ref)
        VKN_TRY(vkn_launch_gather_ref(x, masks_in, d->thr_logit, s.xraw, s.cnt, B, N, C, P, st));
    else if (bits_in)
        VKN_TRY(vkn_launch_gather_bits(x, bits_in, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt, s.status, touch_own, touch_own_bytes));
    else
        VKN_TRY(vkn_launch_gather(x, masks_in, d->thr_logit, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt, s.status, touch_own, touch_own_bytes));

    // folded feat_transform: x_feat = xraw . W_ft^T + cnt (x) b_ft             (:179-180 folded).  With the
    // composite weights x_feat itself is only materialised when the caller asks for it.
    const bool comp = pw.dynft != nullptr;
    float* xfeat = x_feat_out ? x_feat_out : s.xfeat;
    VknEpi e = mk_epi(d);
    if (xfeat_in) {
        xfeat = const_cast<float*>(xfeat_in);
    } else if (has_ft) {
        if (!comp || x_feat_out || need_xfeat) {
            e.bias = w->ft_b; e.rowscale = s.cnt; e.out = xfeat; e.ldo = C;
            VKN_TRY(vkn_launch_gemm(s.xraw, nullptr, C, w->ft_w, pw.ft, M, C, C, 1, nullptr, e, st));
        }
    } else {
        if (hipMemcpyAsync(xfeat, s.xraw, (size_t)M * C * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return VKN_E_LAUNCH;
    }

    // previous_link: the incoming kernels are rewritten from the previous frame's kernels first     knet/video/kernel_update_head.py:324-372
    if (pre_link) {
        PrepW pl;
        VKN_TRY(carve_pw(d, so->link_pre, flags, &pl));
        VKN_TRY(run_link(d, so->link_pre, pl, obj_in, so->prev_pre, s.lobj, s, st, so->link_pre->dyn_w ? xfeat : nullptr, flags));  // (link_atten: no updator)
        obj_in = s.lobj;
    }

    const float* kb = has_ft ? s.kb : nullptr;
    const bool fewrow = comp && chain_ks_ok(d, w, pw, flags, w->fc_cls_w && cls_logits);
    const bool fast = fewrow || (comp && chain_fast_ok(d, w, pw, flags, w->fc_cls_w && cls_logits));
    if (fast) {
        // (ii) + FC branches: few rows — nine column-spread GEMM phases + the attention (vkn_ksplit.hip); many rows — k_chain_a ->
        // attention -> k_chain_c (vkn_chain.hip); obj_out, cls, kb and the decode kernels are final
        const bool raw = !xfeat_in;
        if (fewrow) {
            VKN_TRY(run_chain_ks(d, w, pw, raw ? s.xraw : xfeat, raw, s.cnt, obj_in, obj_out, cls_logits, cls_sigmoid,
                                 ref_decode ? (chain_only ? kern_out : s.kern32) : nullptr, s, st, obj_ready));
        } else {
            VKN_TRY(run_chain_fast(d, w, pw, raw ? s.xraw : xfeat, raw, s.cnt, obj_in, obj_out, cls_logits, cls_sigmoid,
                                   ref_decode ? (chain_only ? kern_out : s.kern32) : nullptr, s, st,
                                   chain_h2(pw, flags, w->fc_cls_w && cls_logits)));
            if (obj_ready && hipEventRecord(obj_ready, st) != hipSuccess) return VKN_E_LAUNCH;
        }
        if (chain_only) {
            if (kb_out && hipMemcpyAsync(kb_out, s.kb, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return VKN_E_LAUNCH;
        } else if (skip_decode) {
        } else if (ref_decode) VKN_TRY(vkn_launch_decode_ref(x, s.kern32, kb, masks_out, B, N, C, P, st));
        else if (gather_out)
            VKN_TRY(vkn_launch_fused_decode_gather(x, s.kfh, s.kfl, kb, d->thr_logit, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt, s.status, touch_nx, touch_nx_bytes));
        else if (bits_out) VKN_TRY(vkn_launch_decode_bits(x, s.kfh, s.kfl, kb, bits_out, d->thr_logit, B, N, C, P, st, xdt));
        else VKN_TRY(decode_final(kb));
        if (prev_obj && track_out) {
            if (so && so->link_track) {
                PrepW pt;
                VKN_TRY(carve_pw(d, so->link_track, flags, &pt));
                VKN_TRY(run_link(d, so->link_track, pt, obj_out, prev_obj, track_out, s, st, so->track_src == 2 ? obj_out : xfeat, flags));
            } else {
                VKN_TRY(run_link(d, w, pw, obj_out, prev_obj, track_out, s, st, nullptr, flags));
            }
        }
        return VKN_OK;
    }

    // (ii-a) KernelUpdator                                    knet/kernel_updator.py:56-93
    VKN_TRY(run_updator(d, w, pw, xfeat, (comp && !xfeat_in) ? s.xraw : nullptr, s.cnt, obj_in, s.obj1, s, st));

    // (ii-b) kernel interaction: MHA + LN, FFN + LN           knet/det/kernel_update_head.py:204-215
    VKN_TRY(run_attention(d, s, s.obj1, s.obj1, d->heads, w->attn_in_w, pw.attn_in, nullptr, w->attn_in_b, w->attn_out_w,
                          pw.attn_out, w->attn_out_b, w->attn_norm_w, w->attn_norm_b, s.obj2, st));
    const float* obj3 = s.obj2;
    if (w->ffn1_w) {
        VKN_TRY(run_ffn(d, s, s.obj2, w->ffn1_w, pw.ffn1, w->ffn1_b, w->ffn2_w, pw.ffn2, w->ffn2_b, w->ffn_norm_w,
                        w->ffn_norm_b, obj_out, st));
        obj3 = obj_out;
    } else {
        if (hipMemcpyAsync(obj_out, s.obj2, (size_t)M * C * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return VKN_E_LAUNCH;
        obj3 = obj_out;
    }

    // obj_out is final here: the fused head forks the tracking link onto its side stream at this point
    const bool fork_late = vkn_dbg_env("VKN_LINK_FORK_LATE", 0) != 0;  // debug A/B: 1 = fork the link behind the decode launch instead of at obj_out (measured: 8777 vs 8832 frames/s; the decode is equally fast either way)
    if (obj_ready && !fork_late && hipEventRecord(obj_ready, st) != hipSuccess) return VKN_E_LAUNCH;

    // cls and mask branches (:217-227) are independent: layer i of both runs as one grouped launch, then fc_cls + fc_mask
    // (the latter also emits the folded decode bias kb = mask_feat . b_ft).
    const float* tc = obj3;
    const float* tm = obj3;
    const int nl = d->n_cls_fcs > d->n_mask_fcs ? d->n_cls_fcs : d->n_mask_fcs;
    for (int i = 0; i < nl; ++i) {
        VknGemmProb pr[2];
        int np = 0;
        if (i < d->n_cls_fcs) {
            float* dst = (i & 1) ? s.t2 : s.t1;
            e = mk_epi(d); e.ln_w = w->cls_ln_w[i]; e.ln_b = w->cls_ln_b[i]; e.act = 1; e.out = dst; e.ldo = C;
            pr[np++] = VknGemmProb{tc, nullptr, nullptr, nullptr, C, w->cls_fc_w[i], pw.cls_fc[i], C, e};
            tc = dst;
        }
        if (i < d->n_mask_fcs) {
            float* dst = (i & 1) ? s.lq : s.f;   // scratch not otherwise live here
            e = mk_epi(d); e.ln_w = w->mask_ln_w[i]; e.ln_b = w->mask_ln_b[i]; e.act = 1; e.out = dst; e.ldo = C;
            if (comp && i == d->n_mask_fcs - 1) { e.dot_vec = pw.dvec; e.dot_bias = pw.kb0; e.dot_out = s.kb; }  // decode bias
            pr[np++] = VknGemmProb{tm, nullptr, nullptr, nullptr, C, w->mask_fc_w[i], pw.mask_fc[i], C, e};
            tm = dst;
        }
        VKN_TRY(vkn_launch_gemm_group(pr, np, M, C, 1, nullptr, st));
    }
    if (comp) {
        // fc_cls, and the decode kernels Kf = fc_mask(.) . W_ft in ONE GEMM from the composite weight          (:221, :227, :247)
        VknGemmProb pr[2];
        int np = 0;
        if (w->fc_cls_w && cls_logits) {  // heads without a classification branch (knet_vis tracker stages with with_cls=False)
            e = mk_epi(d); e.bias = w->fc_cls_b; e.out = cls_logits; e.ldo = d->ncls;
            if (cls_sigmoid) e.act = 2;  // fused head, last stage: the caller wants cls_score.sigmoid() (knet/det/kernel_iter_head.py:307-308)
            pr[np++] = VknGemmProb{tc, nullptr, nullptr, nullptr, C, w->fc_cls_w, pw.fc_cls, d->ncls, e};
        }
        e = mk_epi(d); e.bias = pw.decb; e.ldo = C;
        if (ref_decode) e.out = chain_only ? kern_out : s.kern32;
        else { e.plane_hi = s.kfh; e.plane_lo = s.kfl; e.rows_per_frame = N; e.NPT = npt_of(N); }
        pr[np++] = VknGemmProb{tm, nullptr, nullptr, nullptr, C, pw.dec32, pw.dec, C, e};
        VKN_TRY(vkn_launch_gemm_group(pr, np, M, C, 1, nullptr, st));
        if (chain_only) {
            if (kb_out && hipMemcpyAsync(kb_out, s.kb, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return VKN_E_LAUNCH;
        } else if (skip_decode) {
            // the caller decodes all frames at once (frame-sequential last stage)
        } else if (ref_decode) VKN_TRY(vkn_launch_decode_ref(x, s.kern32, kb, masks_out, B, N, C, P, st));
        else if (gather_out)
            VKN_TRY(vkn_launch_fused_decode_gather(x, s.kfh, s.kfl, kb, d->thr_logit, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt, s.status, touch_nx, touch_nx_bytes));
        else if (bits_out) VKN_TRY(vkn_launch_decode_bits(x, s.kfh, s.kfl, kb, bits_out, d->thr_logit, B, N, C, P, st, xdt));
        else VKN_TRY(decode_final(kb));
    } else {
        {
            VknGemmProb pr[2];
            int np = 0;
            if (w->fc_cls_w && cls_logits) {
                e = mk_epi(d); e.bias = w->fc_cls_b; e.out = cls_logits; e.ldo = d->ncls;
                if (cls_sigmoid) e.act = 2;
                pr[np++] = VknGemmProb{tc, nullptr, nullptr, nullptr, C, w->fc_cls_w, pw.fc_cls, d->ncls, e};
            }
            e = mk_epi(d); e.bias = w->fc_mask_b; e.out = (chain_only && !has_ft) ? kern_out : s.maskfeat; e.ldo = C;
            if (has_ft) { e.dot_vec = w->ft_b; e.dot_out = s.kb; }
            pr[np++] = VknGemmProb{tm, nullptr, nullptr, nullptr, C, w->fc_mask_w, pw.fc_mask, C, e};
            VKN_TRY(vkn_launch_gemm_group(pr, np, M, C, 1, nullptr, st));
        }
        // (iii) mask decode with the folded kernels  Kf = mask_feat . W_ft   :247-260
        if (chain_only) {
            if (has_ft) {
                e = mk_epi(d); e.out = kern_out; e.ldo = C;
                VKN_TRY(vkn_launch_gemm(s.maskfeat, nullptr, C, w->ft_wT, pw.ftT, M, C, C, 1, nullptr, e, st));
                if (kb_out && hipMemcpyAsync(kb_out, s.kb, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return VKN_E_LAUNCH;
            } else if (kb_out && hipMemsetAsync(kb_out, 0, (size_t)M * sizeof(float), st) != hipSuccess) {
                return VKN_E_LAUNCH;
            }
        } else if (ref_decode) {
            const float* kern = s.maskfeat;
            if (has_ft) {
                e = mk_epi(d); e.out = s.kern32; e.ldo = C;
                VKN_TRY(vkn_launch_gemm(s.maskfeat, nullptr, C, w->ft_wT, pw.ftT, M, C, C, 1, nullptr, e, st));
                kern = s.kern32;
            } else if (skip_decode) {
                if (hipMemcpyAsync(s.kern32, s.maskfeat, (size_t)M * C * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return VKN_E_LAUNCH;
            }
            if (!skip_decode) VKN_TRY(vkn_launch_decode_ref(x, kern, kb, masks_out, B, N, C, P, st));
        } else {
            if (has_ft) {
                e = mk_epi(d); e.plane_hi = s.kfh; e.plane_lo = s.kfl; e.ldo = C; e.rows_per_frame = N; e.NPT = npt_of(N);
                VKN_TRY(vkn_launch_gemm(s.maskfeat, nullptr, C, w->ft_wT, pw.ftT, M, C, C, 1, nullptr, e, st));
            } else {
                VKN_TRY(vkn_launch_split_planes(s.maskfeat, s.kfh, s.kfl, B, N, C, st));
            }
            if (skip_decode) {
            } else if (gather_out)
                VKN_TRY(vkn_launch_fused_decode_gather(x, s.kfh, s.kfl, kb, d->thr_logit, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P,
                                                       st, xdt, s.status, touch_nx, touch_nx_bytes));
            else if (bits_out) VKN_TRY(vkn_launch_decode_bits(x, s.kfh, s.kfl, kb, bits_out, d->thr_logit, B, N, C, P, st, xdt));
            else VKN_TRY(decode_final(kb));
        }
    }

    if (obj_ready && fork_late && hipEventRecord(obj_ready, st) != hipSuccess) return VKN_E_LAUNCH;
    if (prev_obj && track_out) {
        if (so && so->link_track) {   // previous_type "update" (updator on x_feat) / "update_obj" (on obj_feat)        :417-476
            PrepW pt;
            VKN_TRY(carve_pw(d, so->link_track, flags, &pt));
            VKN_TRY(run_link(d, so->link_track, pt, obj3, prev_obj, track_out, s, st, so->track_src == 2 ? obj3 : xfeat, flags));
        } else {
            VKN_TRY(run_link(d, w, pw, obj3, prev_obj, track_out, s, st, nullptr, flags));
        }
    }
    return VKN_OK;
}

// the per-row workspace of frame b alone (B = 1 sub-problem of a B-frame call; gather partials are not per-row: untouched)
StageWs frame_ws(const StageWs& s, const VknDims* d, int b) {
    StageWs r = s;
    const size_t R = (size_t)b * d->N, C = d->C;
    r.xraw += R * C; r.cnt += R; r.xfeat += R * C; r.params += R * 2 * C; r.inputf += R * 2 * C; r.ig += R * 2 * C; r.ug += R * C;
    r.f += R * C; r.obj1 += R * C; r.qkv += R * 3 * C; r.ao += R * C; r.obj2 += R * C; r.h += R * d->ff; r.t1 += R * C; r.t2 += R * C;
    r.maskfeat += R * C; r.kb += R; r.kern32 += R * C; r.lq += R * C; r.lkv += R * 2 * C; r.lobj += R * C; r.lupd += R * C; r.lf += R * C;
    r.kfh += (size_t)b * npt_of(d->N) * C; r.kfl += (size_t)b * npt_of(d->N) * C;
    return r;   // (partial: one frame at a time reuses the base)
}

// workspace of the kernel-initialisation pass
struct InitWs {
    _Float16 *ih, *il, *sh, *sl;
    float *part, *cntp, *cnt, *obj, *seg;
    _Float16 *ph, *pl;   // one-pass form (k_init_pass): the 128 plane rows init_kernels | conv_seg (+ 32 rows of slack for the second split)
    unsigned* bits;      // ... and the thing bits [B][P/64][2][npt]
};
size_t carve_init(int B, int Np, int ncls, int C, int P, bool need_seg, char* base, InitWs* s) {
    Carver c{base, 0};
    const size_t G = vkn_gather_groups(B, P), NPTp = npt_of(Np), NPTs = npt_of(ncls > 0 ? ncls : 1);
    s->ih = c.take<_Float16>(NPTp * C);
    s->il = c.take<_Float16>(NPTp * C);
    s->sh = c.take<_Float16>(NPTs * C);
    s->sl = c.take<_Float16>(NPTs * C);
    s->part = c.take<float>((size_t)B * G * NPTp * C);
    s->cntp = c.take<float>((size_t)B * G * NPTp);
    s->cnt = c.take<float>((size_t)B * Np);
    s->obj = c.take<float>((size_t)B * Np * C);
    s->seg = need_seg ? c.take<float>((size_t)B * ncls * P) : nullptr;
    s->ph = s->pl = nullptr;
    s->bits = nullptr;
    if (ncls > 0 && vkn_init_pass_supported(Np, ncls, C, P)) {
        s->ph = c.take<_Float16>((size_t)160 * C);
        s->pl = c.take<_Float16>((size_t)160 * C);
        s->bits = c.take<unsigned>((size_t)B * (P / 32) * NPTp);
    }
    return (c.off + 255) & ~(size_t)255;
}

}  // namespace

extern "C" {

int vkn_version(void) { return VKN_VERSION; }
size_t vkn_sizeof_dims(void) { return sizeof(VknDims); }
size_t vkn_sizeof_stage_weights(void) { return sizeof(VknStageWeights); }

const char* vkn_strerror(int code) {
    switch (code) {
        case VKN_OK: return "ok";
        case VKN_E_ARG: return "invalid argument (null pointer or non-positive size)";
        case VKN_E_SHAPE:
            return "unsupported shape (need C % 32 == 0, C <= 256, N <= 256, head_dim a power of two in [4, 64], ff % 32 == 0, ncls <= 256, "
                   "conv_kernel_size == 1)";
        case VKN_E_WORKSPACE: return "workspace missing or too small";
        case VKN_E_LAUNCH: return "HIP launch failed";
        case VKN_E_ALIGN: return "pointer not 16-byte aligned";
        case VKN_E_RANGE:
            return "feature map outside the f16-split envelope: |x| >= 65504 or a non-finite x reached a mask gather (its sums are not finite)";
        default: return "unknown error";
    }
}

size_t vkn_gather_workspace_bytes(int B, int N, int C, int P) {
    if (B <= 0 || N <= 0 || C <= 0 || P <= 0) return 0;
    const size_t G = vkn_gather_groups(B, P), NPT = npt_of(N);
    Carver c{nullptr, 0};
    c.take<float>((size_t)B * G * NPT * C);
    c.take<float>((size_t)B * G * NPT);
    c.take<float>((size_t)B * N);
    return (c.off + 255) & ~(size_t)255;
}

int vkn_mask_gather_f32(const float* x, const float* mask_logits, float thr_logit, float* xraw_out, float* cnt_out, int B,
                        int N, int C, int P, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    if (!x || !mask_logits || !xraw_out || B <= 0 || N <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(mask_logits) || !aligned16(xraw_out)) return VKN_E_ALIGN;
    if (C % 32 != 0 || C > 256 || N > 256) return VKN_E_SHAPE;
    if (!ws || ws_bytes < vkn_gather_workspace_bytes(B, N, C, P)) return VKN_E_WORKSPACE;
    const size_t G = vkn_gather_groups(B, P), NPT = npt_of(N);
    Carver c{static_cast<char*>(ws), 0};
    float* part = c.take<float>((size_t)B * G * NPT * C);
    float* cntp = c.take<float>((size_t)B * G * NPT);
    float* cnt = c.take<float>((size_t)B * N);
    if (cnt_out) cnt = cnt_out;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & VKN_FLAG_REF_KERNELS) {
        if (xdt_of(flags)) return VKN_E_SHAPE;
        return vkn_launch_gather_ref(x, mask_logits, thr_logit, xraw_out, cnt, B, N, C, P, st);
    }
    return vkn_launch_gather(x, mask_logits, thr_logit, xraw_out, cnt, part, cntp, B, N, C, P, st, xdt_of(flags));
}

int vkn_mask_gather_real_f32(const float* x, const float* a, float* out, float* asum_out, int B, int N, int C, int P, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!x || !a || !out || B <= 0 || N <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(a) || !aligned16(out)) return VKN_E_ALIGN;
    if (C % 32 != 0 || C > 256 || N > 256) return VKN_E_SHAPE;
    if (!ws || ws_bytes < vkn_gather_workspace_bytes(B, N, C, P)) return VKN_E_WORKSPACE;
    const size_t G = vkn_gather_groups(B, P), NPT = npt_of(N);
    Carver c{static_cast<char*>(ws), 0};
    float* part = c.take<float>((size_t)B * G * NPT * C);
    float* cntp = c.take<float>((size_t)B * G * NPT);
    float* asum = c.take<float>((size_t)B * N);
    if (asum_out) asum = asum_out;
    return vkn_launch_gather_real(x, a, out, asum, part, cntp, B, N, C, P, N, static_cast<hipStream_t>(stream));
}

size_t vkn_decode_workspace_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    Carver c{nullptr, 0};
    c.take<_Float16>((size_t)B * npt_of(N) * C);
    c.take<_Float16>((size_t)B * npt_of(N) * C);
    return (c.off + 255) & ~(size_t)255;
}

int vkn_mask_decode_f32(const float* x, const float* kernels, const float* bias, float* out, int B, int N, int C, int P,
                        void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    if (!x || !kernels || !out || B <= 0 || N <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(kernels) || !aligned16(out)) return VKN_E_ALIGN;
    if (C % 32 != 0 || C > 256 || N > 256) return VKN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((flags & VKN_FLAG_REF_KERNELS) || (P & 1)) {
        if (xdt_of(flags)) return VKN_E_SHAPE;
        return vkn_launch_decode_ref(x, kernels, bias, out, B, N, C, P, st);
    }
    if (!ws || ws_bytes < vkn_decode_workspace_bytes(B, N, C)) return VKN_E_WORKSPACE;
    Carver c{static_cast<char*>(ws), 0};
    _Float16* kfh = c.take<_Float16>((size_t)B * npt_of(N) * C);
    _Float16* kfl = c.take<_Float16>((size_t)B * npt_of(N) * C);
    VKN_TRY(vkn_launch_split_planes(kernels, kfh, kfl, B, N, C, st));
    return vkn_launch_decode(x, kfh, kfl, bias, out, B, N, C, P, st, xdt_of(flags));
}

int vkn_mask_decode_scaled_f32(const float* x, const float* kernels, const float* bias, const float* out_scale, float* out, int B,
                               int N, int C, int P, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    if (!x || !kernels || !out || !out_scale || B <= 0 || N <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(kernels) || !aligned16(out)) return VKN_E_ALIGN;
    if (C % 32 != 0 || C > 256 || N > 256) return VKN_E_SHAPE;
    if ((flags & VKN_FLAG_REF_KERNELS) || (P & 1)) return VKN_E_SHAPE;
    if (!ws || ws_bytes < vkn_decode_workspace_bytes(B, N, C)) return VKN_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c{static_cast<char*>(ws), 0};
    _Float16* kfh = c.take<_Float16>((size_t)B * npt_of(N) * C);
    _Float16* kfl = c.take<_Float16>((size_t)B * npt_of(N) * C);
    VKN_TRY(vkn_launch_split_planes(kernels, kfh, kfl, B, N, C, st));
    return vkn_launch_decode_ex(x, kfh, kfl, bias, out, B, N, C, P, 0, N, st, xdt_of(flags), out_scale);
}

int vkn_split_planes_f32(const float* kernels, void* kf_hi, void* kf_lo, int B, int N, int C, void* stream) {
    if (!kernels || !kf_hi || !kf_lo || B <= 0 || N <= 0 || C <= 0) return VKN_E_ARG;
    if (!aligned16(kf_hi) || !aligned16(kf_lo)) return VKN_E_ALIGN;
    return vkn_launch_split_planes(kernels, static_cast<_Float16*>(kf_hi), static_cast<_Float16*>(kf_lo), B, N, C,
                                   static_cast<hipStream_t>(stream));
}

int vkn_mask_decode_planes_f32(const float* x, const void* kf_hi, const void* kf_lo, const float* bias, float* out, int B,
                               int N, int C, int P, void* stream) {
    return vkn_mask_decode_planes_x(x, VKN_X_F32, kf_hi, kf_lo, bias, out, B, N, C, P, stream);
}

int vkn_mask_decode_planes_x(const void* x, int x_dtype, const void* kf_hi, const void* kf_lo, const float* bias, float* out, int B,
                             int N, int C, int P, void* stream) {
    if (!x || !kf_hi || !kf_lo || !out || B <= 0 || N <= 0 || C <= 0 || P <= 0 || x_dtype < 0 || x_dtype > 2) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(kf_hi) || !aligned16(kf_lo) || !aligned16(out)) return VKN_E_ALIGN;
    if (C % 32 != 0 || C > 256 || N > 256) return VKN_E_SHAPE;
    return vkn_launch_decode(static_cast<const float*>(x), static_cast<const _Float16*>(kf_hi), static_cast<const _Float16*>(kf_lo),
                             bias, out, B, N, C, P, static_cast<hipStream_t>(stream), x_dtype);
}

int vkn_decode_gather_supported(int C, int P) { return vkn_fused_supported(C, P); }

int vkn_decode_gather_f32(const float* x, const void* kf_hi, const void* kf_lo, const float* bias, float thr_logit,
                          float* xraw_out, float* cnt_out, int B, int N, int C, int P, void* ws, size_t ws_bytes, void* stream) {
    return vkn_decode_gather_x(x, VKN_X_F32, kf_hi, kf_lo, bias, thr_logit, xraw_out, cnt_out, B, N, C, P, ws, ws_bytes, stream);
}

int vkn_decode_gather_x(const void* xv, int x_dtype, const void* kf_hi, const void* kf_lo, const float* bias, float thr_logit,
                        float* xraw_out, float* cnt_out, int B, int N, int C, int P, void* ws, size_t ws_bytes, void* stream) {
    const float* x = static_cast<const float*>(xv);
    if (x_dtype < 0 || x_dtype > 2) return VKN_E_ARG;
    if (!x || !kf_hi || !kf_lo || !xraw_out || !cnt_out || B <= 0 || N <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(kf_hi) || !aligned16(kf_lo) || !aligned16(xraw_out)) return VKN_E_ALIGN;
    if (N > 256 || !vkn_fused_supported(C, P)) return VKN_E_SHAPE;
    if (!ws || ws_bytes < vkn_gather_workspace_bytes(B, N, C, P)) return VKN_E_WORKSPACE;
    const size_t G = vkn_gather_groups(B, P), NPT = npt_of(N);
    Carver c{static_cast<char*>(ws), 0};
    float* part = c.take<float>((size_t)B * G * NPT * C);
    float* cntp = c.take<float>((size_t)B * G * NPT);
    return vkn_launch_fused_decode_gather(x, static_cast<const _Float16*>(kf_hi), static_cast<const _Float16*>(kf_lo), bias,
                                          thr_logit, xraw_out, cnt_out, part, cntp, B, N, C, P, static_cast<hipStream_t>(stream),
                                          x_dtype);
}

int vkn_track_link_f32(const VknDims* d, const VknStageWeights* w, const float* cur_obj, const float* prev_obj,
                       float* track_out, void* ws, size_t ws_bytes, void* stream) {
    return vkn_track_link_flags_f32(d, w, cur_obj, prev_obj, track_out, ws, ws_bytes, 0u, stream);
}

int vkn_track_link_flags_f32(const VknDims* d, const VknStageWeights* w, const float* cur_obj, const float* prev_obj,
                             float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !cur_obj || !prev_obj || !track_out) return VKN_E_ARG;
    if (!aligned16(cur_obj) || !aligned16(prev_obj) || !aligned16(track_out)) return VKN_E_ALIGN;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    PrepW pw{};
    if (w->prepared) {
        PrepItem items[40];
        if (carve_prepared(d, w, static_cast<char*>(const_cast<void*>(w->prepared)), &pw, items, nullptr) > w->prepared_bytes)
            return VKN_E_WORKSPACE;
    }
    return run_link(d, w, pw, cur_obj, prev_obj, track_out, s, static_cast<hipStream_t>(stream), nullptr, flags);
}

int vkn_upsample_bilinear_f32(const float* in, float* out, int planes, int H, int W, int S, void* stream) {
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0) return VKN_E_ARG;
    if (!aligned16(in) || !aligned16(out)) return VKN_E_ALIGN;
    return vkn_launch_upsample(in, out, planes, H, W, S, static_cast<hipStream_t>(stream));
}

int vkn_upsample_bilinear_f16out(const float* in, void* out_f16, int planes, int H, int W, int S, void* stream) {
    if (!in || !out_f16 || planes <= 0 || H <= 0 || W <= 0) return VKN_E_ARG;
    return vkn_launch_upsample(in, static_cast<float*>(out_f16), planes, H, W, S, static_cast<hipStream_t>(stream), 1);
}

// ---------------------------------------------------------------------------------------------- kernel initialisation
size_t vkn_kernel_init_workspace_bytes(int B, int Np, int ncls, int C, int P) {
    if (B <= 0 || Np <= 0 || C <= 0 || P <= 0 || ncls < 0) return 0;
    InitWs s;
    return carve_init(B, Np, ncls, C, P, true, nullptr, &s);
}

int vkn_kernel_init_f32(const float* loc_feats, const float* sem_feats, const float* init_w, const float* seg_w,
                        const float* seg_b, int num_thing_classes, int cat_stuff, int with_obj, float thr_logit, float* x_feats,
                        float* mask_preds, float* seg_preds, float* proposal_feats, int B, int Np, int ncls, int C, int P,
                        void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    if (!loc_feats || !init_w || !mask_preds || !proposal_feats || B <= 0 || Np <= 0 || C <= 0 || P <= 0) return VKN_E_ARG;
    const bool sem = sem_feats != nullptr;
    if (sem && (!seg_w || ncls <= 0 || !x_feats)) return VKN_E_ARG;
    if (cat_stuff && (!sem || num_thing_classes < 0 || num_thing_classes > ncls)) return VKN_E_ARG;
    if (!aligned16(loc_feats) || !aligned16(sem_feats) || !aligned16(mask_preds) || !aligned16(x_feats) || !aligned16(seg_preds) ||
        !aligned16(proposal_feats) || !aligned16(init_w) || !aligned16(seg_w))
        return VKN_E_ALIGN;
    const int nstuff = cat_stuff ? ncls - num_thing_classes : 0, N = Np + nstuff;
    if (C % 32 != 0 || C > 256 || Np > 256 || ncls > 256) return VKN_E_SHAPE;
    InitWs s;
    const size_t need = carve_init(B, Np, ncls, C, P, sem && !seg_preds, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_init(B, Np, ncls, C, P, sem && !seg_preds, static_cast<char*>(ws), &s);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ref = (flags & VKN_FLAG_REF_KERNELS) != 0, ref_decode = ref || (P & 1);
    // 2-byte feature storage (VKN_FLAG_X_F16 / _BF16, round 4): loc_feats, sem_feats and x_feats point at 2-byte elements; the decodes and
    // the gather read them like the head's passes do, x_feats = half(float(sem) + float(loc))
    const int xdt = xdt_of(flags);
    if (xdt && (ref || (P % 64) != 0 || with_obj == 2)) return VKN_E_SHAPE;
    const size_t xes = xdt ? 2 : 4;

    // ONE pass over loc and sem (k_init_pass): both decodes, x = loc + sem, the stuff rows, the thing bits of the gather below
    const bool one_pass = sem && !ref && !xdt && with_obj != 2 && s.ph && !(flags & VKN_FLAG_INIT_SEPARATE) &&
                          vkn_init_pass_supported(Np, ncls, C, P) != 0;
    const float* xf = loc_feats;
    if (one_pass) {
        VKN_TRY(vkn_launch_split_planes(init_w, s.ph, s.pl, 1, Np, C, st));                                   // rows [0, Np) (+ zeros up to 128)
        VKN_TRY(vkn_launch_split_planes(seg_w, s.ph + (size_t)Np * C, s.pl + (size_t)Np * C, 1, ncls, C, st));  // rows [Np, Np + ncls)
        InitPassArgs a{};
        a.loc = loc_feats; a.sem = sem_feats; a.kh = s.ph; a.kl = s.pl; a.seg_b = seg_b; a.x_out = x_feats; a.masks = mask_preds;
        a.seg = seg_preds; a.bits = with_obj ? s.bits : nullptr; a.thr = thr_logit; a.Np = Np; a.ncls = ncls; a.nth = num_thing_classes;
        a.cat = nstuff > 0 ? 1 : 0; a.N = N; a.C = C; a.P = P; a.npt = npt_of(Np);
        VKN_TRY(vkn_launch_init_pass(a, B, st));
        xf = x_feats;
    } else {
        // mask_preds[:, :Np] = init_kernels(loc_feats): 1x1 conv, no bias, the same kernels for every frame          (:222)
        if (ref_decode) {
            VKN_TRY(vkn_launch_decode_ref_ex(loc_feats, init_w, nullptr, mask_preds, B, Np, C, P, 1, N, st));
        } else {
            VKN_TRY(vkn_launch_split_planes(init_w, s.ih, s.il, 1, Np, C, st));
            VKN_TRY(vkn_launch_decode_ex(loc_feats, s.ih, s.il, nullptr, mask_preds, B, Np, C, P, 1, N, st, xdt));
        }
        if (sem) {
            // seg_preds = conv_seg(semantic_feats)                                                                   (:231-234)
            float* seg = seg_preds ? seg_preds : s.seg;
            if (ref_decode) {
                VKN_TRY(vkn_launch_decode_ref_ex(sem_feats, seg_w, seg_b, seg, B, ncls, C, P, 1, ncls, st));
            } else {
                VKN_TRY(vkn_launch_split_planes(seg_w, s.sh, s.sl, 1, ncls, C, st));
                VKN_TRY(vkn_launch_decode_ex(sem_feats, s.sh, s.sl, seg_b, seg, B, ncls, C, P, 1, ncls, st, xdt));
            }
            // mask_preds[:, Np:] = seg_preds[:, num_thing_classes:]  (cat_stuff_mask, inference)                      (:255-257)
            if (nstuff > 0) {
                if (hipMemcpy2DAsync(mask_preds + (size_t)Np * P, (size_t)N * P * sizeof(float), seg + (size_t)num_thing_classes * P,
                                     (size_t)ncls * P * sizeof(float), (size_t)nstuff * P * sizeof(float), B,
                                     hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return VKN_E_LAUNCH;
            }
            // x_feats = semantic_feats + loc_feats                                                                   (:238-241)
            if (xdt) VKN_TRY(vkn_launch_add2_half(sem_feats, loc_feats, x_feats, (size_t)B * C * P, xdt, st));
            else VKN_TRY(vkn_launch_add2(sem_feats, loc_feats, x_feats, (size_t)B * C * P, st));
            xf = x_feats;
        } else if (x_feats && x_feats != loc_feats) {
            if (hipMemcpyAsync(x_feats, loc_feats, (size_t)B * C * P * xes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return VKN_E_LAUNCH;
        }
    }
    // obj_feats = einsum('bnhw,bchw->bnc', (sigmoid(mask_preds) > 0.5).float(), x_feats)   (use_binary)           (:243-250)
    const float* obj = nullptr;
    if (with_obj) {
        if (with_obj == 2) {  // use_binary=False: weights (sigmoid(z) > 0.5) * sigmoid(z)
            if (ref) return VKN_E_SHAPE;  // no exact-fp32 reference kernel for the soft weights
            VKN_TRY(vkn_launch_gather_soft(xf, mask_preds, thr_logit, s.obj, s.cnt, s.part, s.cntp, B, Np, C, P, N, st));
        } else if (ref) VKN_TRY(vkn_launch_gather_ref_ex(xf, mask_preds, thr_logit, s.obj, s.cnt, B, Np, C, P, N, st));
        else if (one_pass) VKN_TRY(vkn_launch_gather_bits(xf, s.bits, s.obj, s.cnt, s.part, s.cntp, B, Np, C, P, st));
        else VKN_TRY(vkn_launch_gather_ex(xf, mask_preds, thr_logit, s.obj, s.cnt, s.part, s.cntp, B, Np, C, P, N, st, xdt));
        obj = s.obj;
    }
    // proposal_feats = init_kernels.weight (+ obj_feats), then the stuff kernels conv_seg.weight[num_thing:]      (:234-263)
    return vkn_launch_init_finish(init_w, obj, seg_w, proposal_feats, B, Np, N, num_thing_classes, C, st);
}

size_t vkn_sizeof_panoptic_cfg(void) { return sizeof(VknPanopticCfg); }

size_t vkn_panoptic_workspace_bytes(const VknPanopticCfg* cfg, int B, int N) {
    if (!cfg || B <= 0 || N <= 0 || cfg->num_proposals > N || cfg->max_per_img <= 0) return 0;
    return vkn_panoptic_ws_bytes(B, cfg->max_per_img + (N - cfg->num_proposals), cfg->Ho, cfg->Wo);
}

int vkn_panoptic_joint_f32(const VknPanopticCfg* cfg, const float* cls_prob, const float* mask_logits, int B, int N, int ncls,
                           int* panoptic_seg, int* info, int* nseg, int* bbox, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !cls_prob || !mask_logits || !panoptic_seg || !info || !nseg || B <= 0 || N <= 0 || ncls <= 0) return VKN_E_ARG;
    if (!ws || !aligned16(ws)) return VKN_E_WORKSPACE;
    return vkn_launch_panoptic_joint(cfg, cls_prob, mask_logits, B, N, ncls, panoptic_seg, info, nseg, bbox, ws, ws_bytes,
                                     static_cast<hipStream_t>(stream));
}

size_t vkn_prepared_bytes(const VknDims* d, const VknStageWeights* w) {
    if (check_dims(d) != VKN_OK || !w) return 0;
    PrepW pw;
    PrepItem items[40];
    return carve_prepared(d, w, nullptr, &pw, items, nullptr);
}

int vkn_prepare_stage_f32(const VknDims* d, const VknStageWeights* w, void* prepared, size_t bytes, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !prepared) return VKN_E_ARG;
    if (!aligned16(prepared)) return VKN_E_ALIGN;
    PrepW pw;
    PrepItem items[40];
    int n = 0;
    if (carve_prepared(d, w, static_cast<char*>(prepared), &pw, items, &n) > bytes) return VKN_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i)
        VKN_TRY(vkn_launch_split_w3(items[i].src, const_cast<void*>(*items[i].dst), items[i].nout, items[i].k, st));
    if (pw.dynft) {
        // composite weights, exact-fp32 GEMMs (out[m][n] = sum_k A[m][k] W[n][k])
        const int C = d->C;
        auto mm = [&](const float* A, const float* W, float* out, int M, int Nout) {
            VknEpi e{};
            e.eps = d->ln_eps; e.out = out; e.ldo = Nout;
            return vkn_launch_gemm(A, nullptr, C, W, nullptr, M, C, Nout, 1, nullptr, e, st);
        };
        VKN_TRY(mm(w->dyn_w, w->ft_wT, pw.dynft32, 2 * C, C));       // (W_dyn.W_ft)[j][c] = sum_k W_dyn[j][k] W_ft[k][c]
        VKN_TRY(mm(w->dyn_w, w->ft_b, pw.bcnt, 2 * C, 1));           // W_dyn.b_ft
        VKN_TRY(vkn_launch_transpose(w->fc_mask_w, pw.fmT, C, C, st));
        VKN_TRY(mm(w->ft_wT, pw.fmT, pw.dec32, C, C));               // (W_ft^T.W_fm)[c][k] = sum_j W_ft[j][c] W_fm[j][k]
        VKN_TRY(mm(w->ft_wT, w->fc_mask_b, pw.decb, C, 1));          // W_ft^T.b_fm
        VKN_TRY(mm(pw.fmT, w->ft_b, pw.dvec, C, 1));                 // W_fm^T.b_ft
        VKN_TRY(mm(w->fc_mask_b, w->ft_b, pw.kb0, 1, 1));            // b_fm.b_ft
        VKN_TRY(vkn_launch_split_w3(pw.dynft32, const_cast<void*>(pw.dynft), 2 * C, C, st));
        VKN_TRY(vkn_launch_split_w3(pw.dec32, const_cast<void*>(pw.dec), C, C, st));
        if (pw.h2_scale) {
            // the fp16-form images: per matrix its power-of-two scale (one launch, on the device), then the split of W * scale
            const int C2 = d->C, FF = d->ff;
            const float* src[VKN_H2_COUNT] = {pw.dynft32, w->dyn_w, w->inp_w, w->ig_w, w->ug_w, w->fc_w, w->attn_in_w, w->attn_out_w, w->ffn1_w,
                                              w->ffn2_w, w->cls_fc_w[0], w->mask_fc_w[0], w->fc_cls_w, pw.dec32};
            const int nout[VKN_H2_COUNT] = {2 * C2, 2 * C2, 2 * C2, C2, C2, C2, 3 * C2, C2, FF, C2, C2, C2, w->fc_cls_w ? d->ncls : 0, C2};
            const int kk[VKN_H2_COUNT] = {C2, C2, C2, C2, C2, C2, C2, C2, C2, FF, C2, C2, C2, C2};
            if (hipMemsetAsync(pw.h2_scratch, 0, 4 * sizeof(unsigned), st) != hipSuccess) return VKN_E_LAUNCH;
            for (int i = 0; i < VKN_H2_COUNT; ++i) {
                if (!pw.h2[i]) continue;
                float* sc = pw.h2_scale + 8 * i;
                VKN_TRY(vkn_pow2_scale_f32(src[i], (size_t)nout[i] * kk[i], 10, sc, pw.h2_scratch, st));
                VKN_TRY(vkn_launch_split_h2(src[i], const_cast<void*>(pw.h2[i]), nout[i], kk[i], sc, st));
            }
        }
        if (pw.chain_consts) {
            VknChainConsts cc{};
            cc.bcnt = pw.bcnt; cc.dyn_b = w->dyn_b;
            cc.norm_out_w = w->norm_out_w; cc.norm_out_b = w->norm_out_b; cc.inp_b = w->inp_b;
            cc.inorm_out_w = w->inorm_out_w; cc.inorm_out_b = w->inorm_out_b;
            cc.ig_b = w->ig_b; cc.inorm_in_w = w->inorm_in_w; cc.inorm_in_b = w->inorm_in_b;
            cc.ug_b = w->ug_b; cc.norm_in_w = w->norm_in_w; cc.norm_in_b = w->norm_in_b;
            cc.fc_b = w->fc_b; cc.fc_norm_w = w->fc_norm_w; cc.fc_norm_b = w->fc_norm_b; cc.in_b = w->attn_in_b;
            cc.out_b = w->attn_out_b; cc.attn_norm_w = w->attn_norm_w; cc.attn_norm_b = w->attn_norm_b;
            cc.ffn1_b = w->ffn1_b; cc.ffn2_b = w->ffn2_b; cc.ffn_norm_w = w->ffn_norm_w; cc.ffn_norm_b = w->ffn_norm_b;
            cc.cls_ln_w = w->cls_ln_w[0]; cc.cls_ln_b = w->cls_ln_b[0]; cc.mask_ln_w = w->mask_ln_w[0]; cc.mask_ln_b = w->mask_ln_b[0];
            cc.dvec = pw.dvec; cc.fc_cls_b = w->fc_cls_w ? w->fc_cls_b : nullptr; cc.dec_b = pw.decb;
            cc.ff = d->ff; cc.ncls = d->ncls;
            for (int i = 0; i < VKN_H2_COUNT; ++i) cc.h2_inv[i] = (pw.h2_scale && pw.h2[i]) ? pw.h2_scale + 8 * i + 4 : nullptr;
            VKN_TRY(vkn_chain_pack_consts(cc, pw.chain_consts, st));
        }
    }
    return VKN_OK;
}

int vkn_split_weight_f32(const float* W, void* w_split, int Nout, int K, void* stream) {
    if (!W || !w_split) return VKN_E_ARG;
    return vkn_launch_split_w3(W, w_split, Nout, K, static_cast<hipStream_t>(stream));
}

int vkn_split_weight_t_f32(const float* W, void* w_split_t, int Nout, int K, void* stream) {
    if (!W || !w_split_t) return VKN_E_ARG;
    return vkn_launch_split_w3_t(W, w_split_t, Nout, K, static_cast<hipStream_t>(stream));
}

int vkn_linear_f32(const float* A, const float* W, const void* w_split, const float* bias, float* out, int M, int K, int Nout,
                   int act, int ksplit, void* ws, size_t ws_bytes, void* stream) {
    if (!A || !W || !out || M <= 0 || K <= 0 || Nout <= 0) return VKN_E_ARG;
    if (K % 32 != 0) return VKN_E_SHAPE;
    if (ksplit > 1 && (!ws || ws_bytes < (size_t)ksplit * M * Nout * sizeof(float))) return VKN_E_WORKSPACE;
    if (w_split && K == 256 && ksplit <= 1 && M <= 32 * VKN_KS_MAX_ROW_TILES && aligned16(A) && aligned16(out) &&
        act >= 0 && act <= 2) {
        // up to 512 rows (the training chain at 1 - 4 frames per step): the column-spread phase kernel of the few-row chain as a
        // plain GEMM — 32 x 32 output tiles over (row tiles x column blocks) workgroups, each one round trip — instead of one
        // 512-thread workgroup per 32 rows that streams the whole weight image (k_gemm_t3: 15 workgroups at 468 rows)
        VknKsProb p{};
        p.pro.a[0] = A; p.pro.lda[0] = K; p.pro.nsum = 1; p.pro.eps = 1e-5f;
        p.Wsplit = w_split; p.Nout = Nout; p.KT = 8;
        p.epi.bias = bias; p.epi.act = act; p.epi.out = out; p.epi.ldo = Nout;
        const int rc = vkn_launch_gemm_ks(&p, 1, 0, 0, 1, 0, M, static_cast<hipStream_t>(stream));
        if (rc != VKN_E_SHAPE) return rc;
    }
    if (w_split && ksplit > 1 && Nout <= 256 && K % ksplit == 0 && (K / ksplit == 256 || K / ksplit == 512) && M <= 32 * VKN_KS_MAX_ROW_TILES &&
        aligned16(A) && aligned16(out) && aligned16(ws) && act >= 0 && act <= 2) {
        // ... and a longer contraction in chunks of 256 / 512 over blockIdx.z of the same kernel (partial products in `ws`), summed in
        // fixed order with bias and activation by the row epilogue
        VknKsProb p{};
        p.pro.a[0] = A; p.pro.lda[0] = K; p.pro.nsum = 1; p.pro.eps = 1e-5f;
        p.Wsplit = w_split; p.Nout = Nout; p.KT = K / 32;
        p.epi.out = static_cast<float*>(ws); p.epi.ldo = Nout;
        const int rc = vkn_launch_gemm_ks(&p, 1, 0, ksplit, K / ksplit / 256, (long long)M * Nout, M, static_cast<hipStream_t>(stream));
        if (rc == VKN_OK) {
            VknEpi e{};
            e.bias = bias; e.act = act; e.out = out; e.ldo = Nout; e.eps = 1e-5f;
            return vkn_launch_rowepi(static_cast<const float*>(ws), ksplit, M, Nout, e, static_cast<hipStream_t>(stream));
        }
        if (rc != VKN_E_SHAPE) return rc;
    }
    VknEpi e{};
    e.bias = bias; e.act = act; e.out = out; e.ldo = Nout; e.eps = 1e-5f;
    return vkn_launch_gemm(A, nullptr, K, W, w_split, M, K, Nout, ksplit, static_cast<float*>(ws), e,
                           static_cast<hipStream_t>(stream));
}

int vkn_kernel_updator_f32(const VknDims* d, const VknStageWeights* w, const float* update_feature,
                           const float* input_feature, float* out, void* ws, size_t ws_bytes, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !update_feature || !input_feature || !out) return VKN_E_ARG;
    if (!aligned16(update_feature) || !aligned16(input_feature) || !aligned16(out)) return VKN_E_ALIGN;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    PrepW pw{};
    if (w->prepared) {
        PrepItem items[40];
        if (carve_prepared(d, w, static_cast<char*>(const_cast<void*>(w->prepared)), &pw, items, nullptr) > w->prepared_bytes)
            return VKN_E_WORKSPACE;
    }
    return run_updator(d, w, pw, update_feature, nullptr, nullptr, input_feature, out, s, static_cast<hipStream_t>(stream));
}

int vkn_workspace_init(void* ws, size_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < 256) return VKN_E_WORKSPACE;
    return hipMemsetAsync(ws, 0, 256, static_cast<hipStream_t>(stream)) == hipSuccess ? VKN_OK : VKN_E_LAUNCH;
}

int vkn_workspace_status(void* ws, size_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < 256) return VKN_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int word = 0;
    if (hipMemcpyAsync(&word, ws, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return VKN_E_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return VKN_E_LAUNCH;
    if (word == 0) return VKN_OK;
    if (hipMemsetAsync(ws, 0, sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;   // read-and-clear
    // any other bit: the header holds foreign data (an entry point without a header was given this buffer from offset 0: include/vkn.h)
    return (word & VKN_STATUS_RANGE) ? VKN_E_RANGE : VKN_E_WORKSPACE;
}

size_t vkn_stage_workspace_bytes(const VknDims* d) {
    if (check_dims(d) != VKN_OK) return 0;
    StageWs s;
    return carve_stage(d, nullptr, &s);
}

int vkn_stage_forward_f32(const VknDims* d, const VknStageWeights* w, const float* x, const float* obj_in,
                          const float* masks_in, const float* prev_obj, float* cls_logits, float* masks_out, float* obj_out,
                          float* x_feat_out, float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !x || !obj_in || !masks_in || !masks_out || !obj_out) return VKN_E_ARG;
    if (w->fc_cls_w && !cls_logits) return VKN_E_ARG;  // cls_logits may be NULL only for stages without a classification branch
    if (!aligned16(x) || !aligned16(obj_in) || !aligned16(masks_in) || !aligned16(masks_out) || !aligned16(obj_out))
        return VKN_E_ALIGN;
    if (masks_in == masks_out) return VKN_E_ARG;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    return run_stage(d, w, x, obj_in, masks_in, prev_obj, cls_logits, masks_out, obj_out, x_feat_out, track_out, s, flags,
                     static_cast<hipStream_t>(stream));
}

int vkn_stage_forward_link_f32(const VknDims* d, const VknStageWeights* w, const VknStageWeights* link_pre,
                               const VknStageWeights* link_track, int track_src, const float* x, const float* obj_in,
                               const float* masks_in, const float* prev_obj, float* cls_logits, float* masks_out, float* obj_out,
                               float* x_feat_out, float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !x || !obj_in || !masks_in || !masks_out || !obj_out) return VKN_E_ARG;
    if (w->fc_cls_w && !cls_logits) return VKN_E_ARG;
    if (track_src < 0 || track_src > 2 || (link_track && track_src == 0) || ((link_pre || link_track) && !prev_obj)) return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(obj_in) || !aligned16(masks_in) || !aligned16(masks_out) || !aligned16(obj_out) ||
        !aligned16(prev_obj))
        return VKN_E_ALIGN;
    if (masks_in == masks_out) return VKN_E_ARG;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    StageOpts so;
    so.link_pre = link_pre;
    so.prev_pre = link_pre ? prev_obj : nullptr;
    so.link_track = track_out ? link_track : nullptr;
    so.track_src = track_src;
    return run_stage(d, w, x, obj_in, masks_in, prev_obj, cls_logits, masks_out, obj_out, x_feat_out, track_out, s, flags,
                     static_cast<hipStream_t>(stream), nullptr, nullptr, false, false, false, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, 0, 0, nullptr, &so);
}

int vkn_link_block_f32(const VknDims* d, const VknStageWeights* w, const float* update_feature, const float* cur,
                       const float* prev, float* out, void* ws, size_t ws_bytes, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !cur || !prev || !out || ((w->dyn_w != nullptr) != (update_feature != nullptr))) return VKN_E_ARG;
    if (!aligned16(cur) || !aligned16(prev) || !aligned16(out) || !aligned16(update_feature)) return VKN_E_ALIGN;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    PrepW pw;
    VKN_TRY(carve_pw(d, w, 0, &pw));
    return run_link(d, w, pw, cur, prev, out, s, static_cast<hipStream_t>(stream), update_feature);
}

// Clip-level query merge of the VIS heads ('attention' / 'attention_pos'):
//   out = LN2(FFN(LN1(query + MHA8(query + pos, keys + pos, keys))))
// kernel_frame_iter_head.py:142-160 (fusing the per-frame object features) and tracker/kernel_update_head.py:244-263 (fusing the
// per-frame gathers).  d: B = clips, N = queries per clip, ff = the merge FFN's width; keys [B][frames * N][C], key f*N + n takes
// pos[n] (key_pos = query_pos.repeat(1, frames, 1)).  `w` carries the pa_* (query_merge_attn + query_merge_norm) and lffn*
// (query_merge_ffn + query_merge_ffn_norm) members.
namespace {
struct MergeWs { float *qp, *kp, *kv, *kv2; };
size_t carve_merge(const VknDims* d, int frames, char* base, StageWs* s, MergeWs* m) {
    const size_t st = carve_stage(d, base, s);
    Carver c{base ? base + st : nullptr, 0};
    const size_t Mq = (size_t)d->B * d->N, Mk = Mq * frames, C = d->C;
    m->qp = c.take<float>(Mq * C);
    m->kp = c.take<float>(Mk * C);
    m->kv = c.take<float>(Mk * 2 * C);
    m->kv2 = c.take<float>(Mk * 2 * C);
    return st + ((c.off + 255) & ~(size_t)255);
}
}  // namespace

size_t vkn_query_merge_workspace_bytes(const VknDims* d, int num_frames) {
    if (check_dims(d) != VKN_OK || num_frames <= 0) return 0;
    StageWs s;
    MergeWs m;
    return carve_merge(d, num_frames, nullptr, &s, &m);
}

int vkn_query_merge_f32(const VknDims* d, int num_frames, const VknStageWeights* w, const float* query, const float* keys,
                        const float* pos, float* out, void* ws, size_t ws_bytes, void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !query || !keys || !out || num_frames <= 0) return VKN_E_ARG;
    if (!w->pa_in_w || !w->pa_in_b || !w->pa_out_w || !w->pa_norm_w || !w->lffn1_w || !w->lffn2_w || !w->lffn_norm_w) return VKN_E_ARG;
    if (!aligned16(query) || !aligned16(keys) || !aligned16(pos) || !aligned16(out)) return VKN_E_ALIGN;
    StageWs s;
    MergeWs m;
    const size_t need = carve_merge(d, num_frames, nullptr, &s, &m);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_merge(d, num_frames, static_cast<char*>(ws), &s, &m);
    PrepW pw;
    VKN_TRY(carve_pw(d, w, 0, &pw));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int C = d->C, Mq = d->B * d->N, Nk = d->N * num_frames, Mk = d->B * Nk, hd = C / 8;
    const float* qsrc = query;
    const float* ksrc = keys;
    if (pos) {
        VKN_TRY(vkn_launch_add_rows(query, pos, m.qp, (size_t)Mq, C, d->N, st));
        VKN_TRY(vkn_launch_add_rows(keys, pos, m.kp, (size_t)Mk, C, d->N, st));
        qsrc = m.qp;
        ksrc = m.kp;
    }
    VknEpi e = mk_epi(d);
    e.bias = w->pa_in_b; e.out = s.lq; e.ldo = C;
    VKN_TRY(vkn_launch_gemm(qsrc, nullptr, C, w->pa_in_w, pw.pa_in, Mq, C, C, 1, nullptr, e, st));
    // k | v rows of the packed in_proj as ONE [2C]-column GEMM (the prepared tile image covers both); with a position table the
    // k columns come from keys + pos and the v columns from keys: the same GEMM on both operands, each read for its half
    e = mk_epi(d);
    e.bias = w->pa_in_b + C; e.out = m.kv; e.ldo = 2 * C;
    VKN_TRY(vkn_launch_gemm(ksrc, nullptr, C, w->pa_in_w + (size_t)C * C, pw.pa_in_kv, Mk, C, 2 * C, 1, nullptr, e, st));
    const float* vsrc = m.kv + C;
    if (pos) {
        e.out = m.kv2;
        VKN_TRY(vkn_launch_gemm(keys, nullptr, C, w->pa_in_w + (size_t)C * C, pw.pa_in_kv, Mk, C, 2 * C, 1, nullptr, e, st));
        vsrc = m.kv2 + C;
    }
    VKN_TRY(vkn_launch_attn(s.lq, C, m.kv, vsrc, 2 * C, s.ao, C, d->B, d->N, Nk, 8, hd, st));
    e = mk_epi(d);   // the residual is the query WITHOUT its position (mmcv: identity = query before `query + query_pos`)
    e.bias = w->pa_out_b; e.resid = query; e.ldr = C; e.ln_w = w->pa_norm_w; e.ln_b = w->pa_norm_b; e.out = s.t1; e.ldo = C;
    VKN_TRY(vkn_launch_gemm(s.ao, nullptr, C, w->pa_out_w, pw.pa_out, Mq, C, C, 1, nullptr, e, st));
    return run_ffn(d, s, s.t1, w->lffn1_w, pw.lffn1, w->lffn1_b, w->lffn2_w, pw.lffn2, w->lffn2_b, w->lffn_norm_w,
                   w->lffn_norm_b, out, st);
}

int vkn_stage_chain_f32(const VknDims* d, const VknStageWeights* w, const float* x_feat, const float* obj_in, float* cls_logits,
                        float* kernels_out, float* kb_out, float* obj_out, void* ws, size_t ws_bytes, unsigned flags,
                        void* stream) {
    VKN_TRY(check_dims(d));
    if (!w || !x_feat || !obj_in || !kernels_out || !obj_out) return VKN_E_ARG;
    if (!aligned16(x_feat) || !aligned16(obj_in) || !aligned16(kernels_out) || !aligned16(obj_out)) return VKN_E_ALIGN;
    StageWs s;
    const size_t need = carve_stage(d, nullptr, &s);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_stage(d, static_cast<char*>(ws), &s);
    return run_stage(d, w, nullptr, obj_in, nullptr, nullptr, cls_logits, nullptr, obj_out, nullptr, nullptr, s, flags,
                     static_cast<hipStream_t>(stream), nullptr, nullptr, false, false, false, nullptr, nullptr, x_feat, kernels_out,
                     kb_out);
}

// Side stream of the fused head: the tracking link (attention over the previous frame's kernels + FFN, a latency-bound [N x C]
// chain) depends only on the last stage's updated kernels, while the main stream still has that stage's cls / mask branches, the
// mask decode and the x4 upsample (HBM-bound) to run.  Fork at "obj_out final", join before the call returns control of the
// workspace.  One side stream + two events per host thread and device, created on first use, never destroyed (process lifetime).
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int dev = -1;
};
static SideStream* side_stream() {
    thread_local SideStream ss[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream* p = &ss[dev];
    if (p->dev != dev) {
        // highest priority the device offers: the link's small launches must not queue behind the 100 k workgroups of the x4 upsample
        // that runs beside them (round 4 trace: the link's first GEMM took 1.4 ms next to the upsample and the rest of the link
        // finished AFTER it — 25 .. 240 us of tail per call)
        int plo = 0, phi = 0;
        if (hipDeviceGetStreamPriorityRange(&plo, &phi) != hipSuccess) plo = phi = 0;
        if (hipStreamCreateWithPriority(&p->st, hipStreamNonBlocking, phi) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&p->join, hipEventDisableTiming) != hipSuccess) return nullptr;
        p->dev = dev;
    }
    return p;
}

static size_t carve_head(const VknDims* d, char* base, StageWs* s, float** mtmp, float** otmp, float** ctmp, unsigned** bits) {
    const size_t stage_bytes = carve_stage(d, base, s);
    Carver c{base, stage_bytes};
    const size_t M = (size_t)d->B * d->N, P = (size_t)d->H * d->W;
    *mtmp = c.take<float>(M * P);
    *otmp = c.take<float>(M * d->C);
    *ctmp = c.take<float>(M * d->ncls);
    const size_t nw = (size_t)d->B * ((P + 31) / 32) * npt_of(d->N);  // bit-packed stage hand-off, two buffers
    bits[0] = c.take<unsigned>(nw);
    bits[1] = c.take<unsigned>(nw);
    return (c.off + 255) & ~(size_t)255;
}

size_t vkn_head_workspace_bytes(const VknDims* d) {
    if (check_dims(d) != VKN_OK) return 0;
    StageWs s;
    float *a, *b, *c;
    unsigned* bits[2];
    return carve_head(d, nullptr, &s, &a, &b, &c, bits);
}

static int head_forward_impl(const VknDims* d, int num_stages, const VknStageWeights* stages, const VknStageWeights* link_pre,
                             const VknStageWeights* link_track, int track_src, const float* x, const float* proposal_feats,
                             const float* mask_preds_in, const float* prev_obj, float* obj_out, float* cls_prob,
                             float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out, void* ws,
                             size_t ws_bytes, unsigned flags, void* stream, void* ev_decode_start, void* ev_decode_stop);

int vkn_head_forward_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const float* x,
                         const float* proposal_feats, const float* mask_preds_in, const float* prev_obj, float* obj_out,
                         float* cls_prob, float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out,
                         void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    return head_forward_impl(d, num_stages, stages, nullptr, nullptr, 0, x, proposal_feats, mask_preds_in, prev_obj, obj_out,
                             cls_prob, mask_preds_out, scaled_out, upsample_stride, track_out, ws, ws_bytes, flags, stream, nullptr,
                             nullptr);
}

int vkn_head_forward_prof_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const float* x,
                              const float* proposal_feats, const float* mask_preds_in, const float* prev_obj, float* obj_out,
                              float* cls_prob, float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out,
                              void* ws, size_t ws_bytes, unsigned flags, void* stream, void* ev_decode_start,
                              void* ev_decode_stop) {
    return head_forward_impl(d, num_stages, stages, nullptr, nullptr, 0, x, proposal_feats, mask_preds_in, prev_obj, obj_out,
                             cls_prob, mask_preds_out, scaled_out, upsample_stride, track_out, ws, ws_bytes, flags, stream,
                             ev_decode_start, ev_decode_stop);
}

int vkn_head_forward_link_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const VknStageWeights* link_pre,
                              const VknStageWeights* link_track, int track_src, const float* x, const float* proposal_feats,
                              const float* mask_preds_in, const float* prev_obj, float* obj_out, float* cls_prob,
                              float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out, void* ws,
                              size_t ws_bytes, unsigned flags, void* stream) {
    if (track_src < 0 || track_src > 2 || (link_track && track_src == 0)) return VKN_E_ARG;
    return head_forward_impl(d, num_stages, stages, link_pre, link_track, track_src, x, proposal_feats, mask_preds_in, prev_obj,
                             obj_out, cls_prob, mask_preds_out, scaled_out, upsample_stride, track_out, ws, ws_bytes, flags, stream,
                             nullptr, nullptr);
}

static int head_forward_impl(const VknDims* d, int num_stages, const VknStageWeights* stages, const VknStageWeights* link_pre,
                             const VknStageWeights* link_track, int track_src, const float* x, const float* proposal_feats,
                             const float* mask_preds_in, const float* prev_obj, float* obj_out, float* cls_prob,
                             float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out, void* ws,
                             size_t ws_bytes, unsigned flags, void* stream, void* ev_decode_start, void* ev_decode_stop) {
    VKN_TRY(check_dims(d));
    if (num_stages <= 0 || !stages || !x || !proposal_feats || !mask_preds_in || !obj_out || !cls_prob || !mask_preds_out)
        return VKN_E_ARG;
    if (!aligned16(x) || !aligned16(proposal_feats) || !aligned16(mask_preds_in) || !aligned16(obj_out) ||
        !aligned16(mask_preds_out) || (scaled_out && !aligned16(scaled_out)))
        return VKN_E_ALIGN;
    if (mask_preds_in == mask_preds_out || proposal_feats == obj_out) return VKN_E_ARG;
    StageWs s;
    float *mtmp, *otmp, *ctmp;
    unsigned* bits[2];
    const size_t need = carve_head(d, nullptr, &s, &mtmp, &otmp, &ctmp, bits);
    if (!ws || ws_bytes < need || !aligned16(ws)) return VKN_E_WORKSPACE;
    carve_head(d, static_cast<char*>(ws), &s, &mtmp, &otmp, &ctmp, bits);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // stage s -> s+1 hand-off as bit words (the only thing the next gather reads) unless the exact-fp32 kernels are asked for,
    // the spatial size is not a multiple of the 64-px decode tile, or the caller wants the logits path (A/B)
    const bool use_bits = !(flags & (VKN_FLAG_REF_KERNELS | VKN_FLAG_LOGITS_HANDOFF)) && ((d->H * d->W) % 64) == 0;
    // ... and by default not even those: decode(s) and gather(s + 1) are ONE pass over x (vkn_fused.hip); VKN_FLAG_BITS_HANDOFF
    // keeps the two-kernel bit-word path (A/B; bit-identical results)
    const bool use_fused = use_bits && !(flags & VKN_FLAG_BITS_HANDOFF) && vkn_fused_supported(d->C, d->H * d->W);

    // Phases of a previous_link clip call (VKN_FLAG_PHASE_*; 0 = everything): the frames of a clip sharded over ranks need the
    // previous rank's final kernels BEFORE their own last-stage chains and should hand their own on BEFORE the HBM-bound tail —
    //   A: stages 0 .. S-2 and the last stage's gather (no cross-frame dependency),
    //   B: the last stage's [N x C] chains, frame by frame (frame 0 links to `prev_obj`),
    //   C: the batched last decode, the upsample and the tracking link.
    // The state between phases (gather sums, stage S-2's kernels) lives in `ws` and in the caller's output tensors.
    const unsigned ph = flags & VKN_FLAG_PHASE_MASK;
    const bool do_a = !ph || (ph & VKN_FLAG_PHASE_A), do_b = !ph || (ph & VKN_FLAG_PHASE_B), do_c = !ph || (ph & VKN_FLAG_PHASE_C);
    if (ph && !(link_pre && (flags & VKN_FLAG_CLIP_LINK) && prev_obj)) return VKN_E_ARG;   // phases exist for the frame-sequential path only

    const float* m_in = mask_preds_in;
    const float* o_in = proposal_feats;
    SideStream* joined = nullptr;
    bool up_done = false;
    for (int sidx = 0; sidx < num_stages; ++sidx) {
        const bool last = (sidx == num_stages - 1);
        if (!last && !do_a) {   // (a later phase: the stage ran in phase A; follow the buffer alternation only)
            const bool to_out_s = ((num_stages - 1 - sidx) & 1) == 0;
            m_in = to_out_s ? mask_preds_out : mtmp;
            o_in = to_out_s ? obj_out : otmp;
            continue;
        }
        // alternate so that the LAST stage writes the caller's buffers
        const bool to_out = ((num_stages - 1 - sidx) & 1) == 0;
        float* m_out = to_out ? mask_preds_out : mtmp;
        float* o_out = to_out ? obj_out : otmp;
        const float* prev = (last && track_out) ? prev_obj : nullptr;   // knet/video/kernel_iter_head.py:544-546
        const bool clip = prev && (flags & VKN_FLAG_CLIP_LINK);
        // the link runs on the side stream (forked where obj_out is final) unless the caller asks for one stream
        SideStream* side = (prev && !(flags & VKN_FLAG_SERIAL_LINK)) ? side_stream() : nullptr;
        const bool link_after = clip || side;
        const float* prev_in_stage = link_after ? nullptr : prev;
        const unsigned* b_in = (use_bits && !use_fused && sidx > 0) ? bits[(sidx - 1) & 1] : nullptr;
        unsigned* b_out = (use_bits && !use_fused && !last) ? bits[sidx & 1] : nullptr;
        // previous_link heads: the last stage's incoming kernels are rewritten from the previous frame's FINAL kernels
        // (knet/video/kernel_update_head.py:324-348), so with consecutive frames in one call (clip mode) the last stage's [N x C]
        // chain is frame-sequential: gather (all frames) -> per frame { link, update, interaction, FC branches } -> decode (all
        // frames).  Everything that streams x stays batched; only ~15 small launches per frame are serialised.
        const float* prev_pre = (last && link_pre) ? prev_obj : nullptr;
        const bool seq = prev_pre && (flags & VKN_FLAG_CLIP_LINK) && (d->B > 1 || ph);
        StageOpts so;
        so.link_pre = prev_pre ? link_pre : nullptr;
        so.prev_pre = prev_pre;
        so.link_track = (last && track_out) ? link_track : nullptr;
        so.track_src = track_src;
        if (!last && use_fused && !(flags & (VKN_FLAG_CHAIN_LAUNCHES | VKN_FLAG_EXACT_GEMM)) && stages[sidx + 1].prepared) {
            // the fused pass of this stage ends in the next stage's gather reduction: it warms the weight images the persistent kernels
            // will stream — the fp16 images alone where those run (they sit together in the prepared buffer)
            PrepW pwn;
            PrepItem itn[40];
            carve_prepared(d, &stages[sidx + 1], static_cast<char*>(const_cast<void*>(stages[sidx + 1].prepared)), &pwn, itn, nullptr);
            if ((flags & VKN_FLAG_CHAIN_PERSISTENT) || (d->B * d->N + 31) / 32 >= persistent_min_row_tiles(pwn, flags)) {
                if (chain_h2(pwn, flags, false)) {
                    so.touch_next = pwn.h2[VKN_H2_DYNFT];
                    so.touch_next_bytes = (size_t)(reinterpret_cast<const char*>(pwn.h2_scale) - static_cast<const char*>(pwn.h2[VKN_H2_DYNFT]));
                } else {
                    so.touch_next = stages[sidx + 1].prepared;
                    so.touch_next_bytes = stages[sidx + 1].prepared_bytes;
                }
            }
        }
        hipEvent_t ev0 = last ? static_cast<hipEvent_t>(ev_decode_start) : nullptr;
        hipEvent_t ev1 = last ? static_cast<hipEvent_t>(ev_decode_stop) : nullptr;
        float* up_out = (last && scaled_out && upsample_stride > 1) ? scaled_out : nullptr;
        if (!seq) {
            // the last stage's fc_cls epilogue applies the sigmoid and writes the caller's cls_prob directly
            VKN_TRY(run_stage(d, &stages[sidx], x, o_in, m_in, prev_in_stage, last ? cls_prob : ctmp, m_out, o_out, nullptr,
                              prev_in_stage ? track_out : nullptr, s, flags, st, b_in, b_out, last, use_fused && sidx > 0,
                              use_fused && !last, ev0, ev1, nullptr, nullptr, nullptr, (prev && side) ? side->fork : nullptr, up_out,
                              upsample_stride, vkn_dbg_env("VKN_LAST_CHUNK", 0), &up_done, &so));
        } else {
            const VknStageWeights* w = &stages[sidx];
            const int B = d->B, N = d->N, C = d->C, P = d->H * d->W;
            if (do_a && !(use_fused && sidx > 0)) {  // the stage's gather for all frames (run_stage's step (i))
                if (flags & VKN_FLAG_REF_KERNELS) VKN_TRY(vkn_launch_gather_ref(x, m_in, d->thr_logit, s.xraw, s.cnt, B, N, C, P, st));
                else if (b_in) VKN_TRY(vkn_launch_gather_bits(x, b_in, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt_of(flags), s.status));
                else VKN_TRY(vkn_launch_gather(x, m_in, d->thr_logit, s.xraw, s.cnt, s.part, s.cntp, B, N, C, P, st, xdt_of(flags), s.status));
            }
            VknDims d1 = *d;
            d1.B = 1;
            so.skip_decode = true;
            so.keep_xfeat = so.link_track && so.track_src == 1;   // ... and reads every frame's x_feat ("update") from the workspace:
            so.link_track = nullptr;  // the tracking link runs batched behind the loop (every frame's kernels are known then)
            for (int b = 0; b < (do_b ? B : 0); ++b) {
                const size_t r = (size_t)b * N;
                const StageWs sb = frame_ws(s, d, b);
                so.prev_pre = b == 0 ? prev_obj : o_out + (r - N) * C;
                VKN_TRY(run_stage(&d1, w, x, o_in + r * C, nullptr, nullptr, cls_prob + r * d->ncls, nullptr, o_out + r * C, nullptr,
                                  nullptr, sb, flags, st, nullptr, nullptr, true, true, false, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, 0, 0, nullptr, &so));
            }
            if (!do_c) return VKN_OK;   // phases A / B end here: nothing was forked, nothing to join
            if (prev && side && hipEventRecord(side->fork, st) != hipSuccess) return VKN_E_LAUNCH;
            VKN_TRY(final_decode(d, x, s, w->ft_w ? s.kb : nullptr, m_out, flags, st, ev0, ev1, up_out, upsample_stride,
                                 vkn_dbg_env("VKN_LAST_CHUNK", 0), &up_done));
            so.link_track = (last && track_out) ? link_track : nullptr;
        }
        m_in = m_out;
        o_in = o_out;
        if (prev && link_after) {
            hipStream_t ls = side ? side->st : st;
            if (side && hipStreamWaitEvent(side->st, side->fork, 0) != hipSuccess) return VKN_E_LAUNCH;
            // scratch of the link while the main stream is still inside the stage: the cls / mask branches own t1 / t2 / lq / f,
            // so the link's q / kv projections live in the (finished) self-attention's qkv buffer and its attention output in obj1
            StageWs sl = s;
            if (side) {
                sl.lq = s.qkv;
                sl.lkv = s.qkv + (size_t)d->B * d->N * d->C;
                sl.t1 = s.obj1;
            }
            const float* pv = prev;
            if (clip) {
                // clip mode: prev[0] = the caller's previous-frame kernels, prev[b] = this call's frame b - 1          
                const size_t fr = (size_t)d->N * d->C;
                float* pvb = otmp;  // [B][N][C] scratch: the last stage wrote the caller's obj_out and has read its obj_in (otmp)
                if (hipMemcpyAsync(pvb, prev_obj, fr * sizeof(float), hipMemcpyDeviceToDevice, ls) != hipSuccess) return VKN_E_LAUNCH;
                if (d->B > 1 &&
                    hipMemcpyAsync(pvb + fr, obj_out, (size_t)(d->B - 1) * fr * sizeof(float), hipMemcpyDeviceToDevice, ls) != hipSuccess)
                    return VKN_E_LAUNCH;
                pv = pvb;
            }
            PrepW pw{};
            const VknStageWeights* w = so.link_track ? so.link_track : &stages[sidx];
            VKN_TRY(carve_pw(d, w, flags, &pw));
            // previous_type "update": the link's own KernelUpdator turns (x_feat, previous kernels) into the keys / values (:417-445);
            // s.xfeat of the last stage is still intact (nothing after the stage's feat-transform GEMM writes it)
            VKN_TRY(run_link(d, w, pw, obj_out, pv, track_out, sl, ls, so.link_track ? (track_src == 2 ? obj_out : s.xfeat) : nullptr, flags));
            if (side) {
                if (hipEventRecord(side->join, side->st) != hipSuccess) return VKN_E_LAUNCH;
                joined = side;
            }
        }
    }
    // Where the side-stream link joins the caller's stream.  Default: BEHIND the upsample — in back-to-back calls the link's tail (its small
    // launches queue behind the upsample's 100 k workgroups whatever the stream priority: r05 trace, one link GEMM 1.5 ms beside the
    // upsample, FFN + LayerNorm ~100 us after it) overlaps the next call's first kernels.  VKN_FLAG_JOIN_EARLY joins BEFORE the upsample
    // (the link then overlaps the cls / mask branches and the decode only, the upsample has the chip to itself): measured 1-3 % SLOWER in
    // throughput at 1 .. 16 frames per call, equal at 32 (tools/perf_r05.py --what join, docs/LAB_NOTEBOOK.md) — for callers that need the
    // tracking embeddings of ONE call as early as its masks.  Either way everything the call produced (and every use of the workspace) is
    // ordered before later work on the caller's stream.
    const bool join_early = (flags & VKN_FLAG_JOIN_EARLY) != 0;
    if (joined && join_early && hipStreamWaitEvent(st, joined->join, 0) != hipSuccess) return VKN_E_LAUNCH;
    if (scaled_out && upsample_stride > 1 && !up_done)                                    // :122-130
        VKN_TRY(vkn_launch_upsample(mask_preds_out, scaled_out, d->B * d->N, d->H, d->W, upsample_stride, st, (flags & VKN_FLAG_SCALED_F16) ? 1 : 0));
    if (joined && !join_early && hipStreamWaitEvent(st, joined->join, 0) != hipSuccess) return VKN_E_LAUNCH;
    return VKN_OK;
}

}  // extern "C"
