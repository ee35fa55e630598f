// vkn_launch.h — internal host-side launcher prototypes shared by the .hip translation units of libvkn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// Row-wise epilogue descriptor of k_gemm / k_rowepi (vkn_update.hip).
struct VknEpi {
    const float* bias;      // [Nout] or null
    const float* rowscale;  // [M] or null: bias is multiplied by rowscale[row] (pixel count of the folded feat_transform bias)
    const float* bias2;     // [Nout] or null: second, unscaled bias (used together with a rowscale'd `bias`)
    const float* resid;     // [M][ldr] or null, added before LN
    int ldr;
    const float* ln_w;  // [ncols of a tile] or null -> LayerNorm over the row (of the column tile)
    const float* ln_b;
    int ln_from_col;    // LayerNorm only in column tiles with n0 >= ln_from_col (weights indexed from there); 0 = whole row
    float eps;
    int act;     // 0 none, 1 relu, 2 sigmoid
    float* out;  // [M][ldo] or null
    int ldo;
    const float* dot_vec;   // [Nout] or null: dot_out[row] = sum_col result(row,col) * dot_vec[col] (+ *dot_bias)
    const float* dot_bias;  // device scalar or null
    float* dot_out;
    _Float16* plane_hi;  // or null: f16 split planes [B][NPT][ldo]; row r = b*N+n -> plane row b*NPT+n
    _Float16* plane_lo;
    int rows_per_frame;  // N
    int NPT;
};

// One GEMM problem out = epi(A (.*A2) . W^T) for vkn_launch_gemm_group
struct VknGemmProb {
    const float* A;
    const float* A2;  // or null: elementwise factor of A (same lda)
    const float* A3;  // or null: second product term, the effective operand is A*A2 + A3*A4 (same lda)
    const float* A4;
    int lda;
    const float* W;      // fp32 [Nout][K]
    const void* Wsplit;  // or null: bf16x3 planes of W (k_split_w3)
    int Nout;
    VknEpi epi;
};

int vkn_gather_groups(int B, int P);
// `status` (last argument of the gather launchers that end in k_gather_reduce; NULL = no check): a device int, VKN_STATUS_RANGE is OR-ed
// in when a gathered sum is not finite (include/vkn.h: VKN_E_RANGE).  `touch` / `touch_bytes` (or NULL): memory the reduction pulls into the
// memory-side cache on the way — the weights of the chain that follows (k_gather_reduce)
int vkn_launch_gather_ex(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp,
                         int B, int N, int C, int P, int mask_rows, hipStream_t stream, int xdt = 0, int* status = nullptr,
                         const void* touch = nullptr, size_t touch_bytes = 0);
int vkn_launch_gather_ref_ex(const float* x, const float* masks, float thr, float* xraw, float* cnt, int B, int N, int C,
                             int P, int mask_rows, hipStream_t stream);
int vkn_launch_gather_real(const float* x, const float* a, float* xraw, float* cnt, float* part, float* cntp, int B, int N, int C,
                           int P, int mask_rows, hipStream_t stream, int x_alias = -1, float x_floor = 0.f);
int vkn_launch_gather_soft(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp, int B,
                           int N, int C, int P, int mask_rows, hipStream_t stream);
int vkn_launch_gather_bits(const float* x, const unsigned* bits, float* xraw, float* cnt, float* part, float* cntp, int B, int N,
                           int C, int P, hipStream_t stream, int xdt = 0, int* status = nullptr, const void* touch = nullptr,
                           size_t touch_bytes = 0);
int vkn_launch_gather_reduce(const float* part, const float* cntp, float* xraw, float* cnt, int B, int N, int C, int G,
                             hipStream_t stream, int* status = nullptr, const void* touch = nullptr, size_t touch_bytes = 0);
// stage s decode fused with the stage s + 1 gather (vkn_fused.hip)
int vkn_fused_supported(int C, int P);
int vkn_launch_fused_decode_gather(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float thr,
                                   float* xraw, float* cnt, float* part, float* cntp, int B, int N, int C, int P,
                                   hipStream_t stream, int xdt = 0, int* status = nullptr, const void* touch = nullptr,
                                   size_t touch_bytes = 0);
int vkn_launch_gather(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp,
                      int B, int N, int C, int P, hipStream_t stream, int xdt = 0, int* status = nullptr, const void* touch = nullptr,
                      size_t touch_bytes = 0);
int vkn_launch_gather_ref(const float* x, const float* masks, float thr, float* xraw, float* cnt, int B, int N, int C,
                          int P, hipStream_t stream);
// per-frame element strides of the decode operands (shared kernels: 0)
struct VknDecodeStrides {
    long long plane;  // kernel planes (f16 elements) / fp32 kernels (ref kernel) per frame
    long long kb;     // bias elements per frame
    long long out;    // output elements per frame
};
// xdt (last argument of the x-streaming launchers): storage type of x — 0 fp32, 1 fp16, 2 bf16 (VKN_X_* in include/vkn.h); for the
// half types `x` points at 2-byte elements
int vkn_launch_decode_ex(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B, int N,
                         int C, int P, int shared, int out_rows, hipStream_t stream, int xdt = 0, const float* oscale = nullptr);
int vkn_launch_decode_ref_ex(const float* x, const float* kern, const float* kb, float* out, int B, int N, int C, int P,
                             int shared, int out_rows, hipStream_t stream);
int vkn_launch_decode_bits(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, unsigned* bits_out,
                           float thr, int B, int N, int C, int P, hipStream_t stream, int xdt = 0);
int vkn_launch_decode(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B, int N,
                      int C, int P, hipStream_t stream, int xdt = 0);
int vkn_launch_decode_ref(const float* x, const float* kern, const float* kb, float* out, int B, int N, int C, int P,
                          hipStream_t stream);
int vkn_launch_split_planes(const float* kern, _Float16* kfh, _Float16* kfl, int B, int N, int C, hipStream_t stream);
int vkn_launch_gemm(const float* A, const float* A2, int lda, const float* W, const void* Wsplit, int M, int K, int Nout,
                    int ksplit, float* partial, const VknEpi& epi, hipStream_t stream);
int vkn_launch_gemm_group(const VknGemmProb* probs, int nprob, int M, int K, int ksplit, float* partial, hipStream_t stream);
size_t vkn_split_w3_bytes(int Nout, int K);
int vkn_launch_add2(const float* a, const float* b, float* out, size_t n, hipStream_t st);
int vkn_launch_add2_half(const void* a, const void* b, void* out, size_t n, int xdt, hipStream_t st);  // 2-byte features (xdt 1 fp16, 2 bf16)
int vkn_launch_add_rows(const float* a, const float* pos, float* out, size_t rows, int C, int period, hipStream_t st);
// pass 0 in one pass over loc and sem (vkn_init.hip: k_init_pass)
struct InitPassArgs {
    const float *loc, *sem;
    const _Float16 *kh, *kl;   // planes [128][C]: rows [0, Np) init_kernels, [Np, Np + ncls) conv_seg (f16 hi / lo split); the rest is not read
    const float* seg_b;        // [ncls] or NULL
    float *x_out, *masks, *seg;   // x [B][C][P]; mask_preds [B][N][P]; seg_preds [B][ncls][P] or NULL
    unsigned* bits;            // [B][P/64][2][npt] thing bits (z >= thr) or NULL
    float thr;
    int Np, ncls, nth, cat, N, C, P, px_per_wg;
    int npt;                   // row stride of the bit words = roundup(Np, 32)
};
int vkn_init_pass_supported(int Np, int ncls, int C, int P);
int vkn_launch_init_pass(const InitPassArgs& a, int B, hipStream_t st);
int vkn_launch_init_finish(const float* init_w, const float* obj, const float* seg_w, float* out, int B, int Np, int N, int nth,
                           int C, hipStream_t st);
int vkn_launch_ffn_fused(const float* X, int ldx, const void* W1s, const float* b1, const void* W2s, int M, int C, int FF, int HS,
                         float* partial, const VknEpi& epi2, hipStream_t stream);
int vkn_launch_transpose(const float* src, float* dst, int R, int Cc, hipStream_t st);  // dst[c][r] = src[r][c]
int vkn_launch_split_w3(const float* W, void* Wp, int Nout, int K, hipStream_t stream);
int vkn_launch_split_w3_t(const float* W, void* Wp, int Nout, int K, hipStream_t stream);  // images of W^T, W stored [K][Nout]
int vkn_launch_ku_mix(const float* params, const float* inputf, const float* ig, const float* ug, const float* no_w,
                      const float* no_b, const float* ino_w, const float* ino_b, float eps, float* f, int M, int C,
                      hipStream_t stream);
int vkn_launch_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* out, int ldo, int B, int Nq,
                    int Nk, int heads, int hd, hipStream_t stream);
int vkn_launch_upsample(const float* in, float* out, int planes, int H, int W, int S, hipStream_t stream, int out_f16 = 0);   // out_f16: `out` points at fp16 elements

// post-head joint panoptic merge (vkn_panoptic.hip); VknPanopticCfg is declared in include/vkn.h
struct VknPanopticCfg;
size_t vkn_panoptic_ws_bytes(int B, int K, int Ho, int Wo);
int vkn_launch_panoptic_joint(const VknPanopticCfg* c, const float* cls, const float* masks, int B, int N, int ncls,
                              int* panoptic_seg, int* info, int* nseg, int* bbox, void* ws, size_t ws_bytes, hipStream_t st);

// ---- persistent row-owner chain kernels (vkn_chain.hip): C == 256, pre-split weights.  `off_*` are byte offsets of tile images
// inside the prepared weight buffer `wbase` (vkn_prepare_stage_f32); `consts` is the packed constant block (vkn_chain_pack_consts).
struct VknChainConsts {   // every bias / LayerNorm vector the two kernels read (NULL: zeros / ones)
    const float *bcnt, *dyn_b;                            // [512]: W_dyn.b_ft (scaled by the pixel count), dynamic_layer bias
    const float *norm_out_w, *norm_out_b, *inp_b, *inorm_out_w, *inorm_out_b;
    const float *ig_b, *inorm_in_w, *inorm_in_b, *ug_b, *norm_in_w, *norm_in_b;
    const float *fc_b, *fc_norm_w, *fc_norm_b, *in_b;     // in_b [768]
    const float *out_b, *attn_norm_w, *attn_norm_b, *ffn1_b, *ffn2_b, *ffn_norm_w, *ffn_norm_b;
    const float *cls_ln_w, *cls_ln_b, *mask_ln_w, *mask_ln_b, *dvec, *fc_cls_b, *dec_b;
    int ff, ncls;
    // fp16-form weight images (vkn_chain_h2.hip): DEVICE scalars 1 / scale of the images in the order of VKN_H2_* (NULL: 1)
    const float* h2_inv[14];
};
// the weight matrices of the persistent chain, in the order of VknChainConsts::h2_inv and PrepW::h2
enum { VKN_H2_DYNFT = 0, VKN_H2_DYN, VKN_H2_INP, VKN_H2_IG, VKN_H2_UG, VKN_H2_FC, VKN_H2_IN, VKN_H2_OUT, VKN_H2_FFN1, VKN_H2_FFN2,
       VKN_H2_CLSFC, VKN_H2_MASKFC, VKN_H2_FCCLS, VKN_H2_DEC, VKN_H2_COUNT };
size_t vkn_chain_consts_floats();
int vkn_chain_pack_consts(const VknChainConsts& c, float* out, hipStream_t stream);
struct VknChainA {   // KernelUpdator + attention in_proj
    const float* a0;        // [M][256] update feature (raw gather with composite weights, else x_feat)
    const float* obj_in;    // [M][256]
    const float* rowscale;  // [M] pixel counts: a0 is the RAW gather (dynamic bias = bcnt x count + dyn_b); NULL: a0 is x_feat
    const void* wbase;
    size_t wbytes;
    unsigned off_dyn, off_inp, off_ig, off_ug, off_fc, off_in;
    const float* consts;
    float eps;
    int M;
    float* obj1;  // [M][256]
    float* qkv;   // [M][768]
    int* status;  // the workspace's status word or NULL: the two-term fp16 form ORs VKN_STATUS_RANGE in when an unscaled activation image
                  // (LayerNorm outputs, FFN hidden rows, branch inputs) holds |v| >= 2^15 — one binade below where fp16 turns it into inf
};
struct VknChainC {   // attention out_proj + LN, FFN + LN, cls / mask FCs, fc_cls, folded decode kernels
    const float* ao;      // [M][256]
    const float* obj1;    // [M][256]
    const void* wbase;
    size_t wbytes;
    unsigned off_out, off_ffn1, off_ffn2, off_clsfc, off_maskfc, off_fccls, off_dec;
    const float* consts;
    const float* kb0;     // device scalar b_fm . b_ft
    int ff, ncls, cls_sigmoid;
    float eps;
    int M;
    float* obj_out;   // [M][256]
    float* cls_out;   // [M][ncls] or NULL (no classification branch)
    float* kb_out;    // [M]
    _Float16 *plane_hi, *plane_lo;   // f16 planes [B][NPT][256] ...
    float* kern_out;                 // ... or fp32 [M][256] (exactly one of the two forms)
    int rows_per_frame, NPT;
    int* status;                     // as VknChainA::status
};
// one GEMM (or two grouped ones) per launch on the chain kernels' register-streaming engine: K == 256, pre-split weights, no split-K
// (vkn_chain.hip: k_gemm_t3); VKN_E_SHAPE = not applicable, take k_gemm_s3
int vkn_launch_gemm_t3(const VknGemmProb* probs, int nprob, int M, int K, hipStream_t stream);
int vkn_launch_chain_a(const VknChainA& p, hipStream_t stream);
int vkn_launch_chain_c(const VknChainC& p, hipStream_t stream);
// the same two kernels on the two-term fp16 split (vkn_chain_h2.hip): `off_*` then name the fp16 tile images (vkn_launch_split_h2:
// W times a power of two per matrix, whose inverse sits in the constant block)
int vkn_launch_chain_a_h2(const VknChainA& p, hipStream_t stream);
int vkn_launch_chain_c_h2(const VknChainC& p, hipStream_t stream);
size_t vkn_split_h2_bytes(int Nout, int K);
int vkn_launch_split_h2(const float* W, void* images, int Nout, int K, const float* scale, hipStream_t stream);
// vkn_pow2_scale_f32 (vkn_loss.hip) is the C-ABI entry point; the prepare step calls it directly

// ---- few-row chain (vkn_ksplit.hip): one GEMM phase per launch, column blocks over the chip, the contraction over the waves of a
// workgroup; row-wise normalisation in the CONSUMER's prologue.  K = 256 per problem (or z-split chunks of 256 kpw).
struct VknKsPro {            // how the A operand [M][256] of a phase is formed from what the previous phases stored
    const float* a[4];       // MODE 0: a[0] (+ s * sum_stride, s < nsum); MODE 1: a[0] .* a[1]; MODE 2: gates a[0] (input), a[1] (update), a[2] = param_out, a[3] = input_out
    int lda[4];
    int nsum;                // MODE 0: number of summands (split-K partials of the producer), 1..4
    long long sum_stride;    // floats between summands
    const float* pbias;      // [256] or null, added before the LayerNorm
    const float* presid;     // [M][ldr] or null, added before the LayerNorm
    int ldr;
    const float* ln_w[4];    // MODE 0: [0] or null; MODE 2: the LayerNorms of a[0..3]
    const float* ln_b[4];
    float eps;
    int act;                 // MODE 0: 1 = ReLU after the LayerNorm
    float* side_out;         // [M][ld_side] or null: the finished A rows are ALSO a result (written by column group 0)
    int ld_side;
    const float* dot_vec;    // [256]: dot_out[row] = A[row] . dot_vec (+ *dot_bias)   (written by column group 0)
    const float* dot_bias;
    float* dot_out;
};
struct VknKsEpi {
    const float* bias;       // [Nout] or null
    const float* rowscale;   // [M] or null: bias is multiplied by rowscale[row]
    const float* bias2;      // [Nout] or null (unscaled)
    const float* resid;      // [M][ldr] or null
    int ldr;
    int act;                 // 0 none, 1 relu, 2 sigmoid
    float* out;              // [M][ldo] or null
    int ldo;
    _Float16* plane_hi;      // or null: f16 split planes [B][NPT][ldo]
    _Float16* plane_lo;
    int rows_per_frame, NPT;
};
struct VknKsProb {
    VknKsPro pro;
    const void* Wsplit;      // bf16x3 tile images of W [Nout][K]
    int Nout;
    int KT;                  // K / 32 of the weight (8; the z-split FFN second Linear: ff / 32)
    VknKsEpi epi;
};
int vkn_launch_rowepi(const float* partial, int ks, int M, int Nout, const VknEpi& epi, hipStream_t stream);
int vkn_launch_gemm_ks(const VknKsProb* probs, int nprob, int mode, int zchunks, int kpw, long long out_zstride, int M, hipStream_t stream);
