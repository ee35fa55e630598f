/* vkn.h — C ABI of libvkn.so: MI355X (gfx950) kernels for Video K-Net's kernel-update head.
 *
 * The reference (lxtGH/Video-K-Net) is 100 % Python and has NO FFI of its own; its boundary for this path is
 * the call signature of three Python methods.  Each entry point below states the reference interface it replaces (file:line,
 * relative to the upstream repo).  The reference-side binding is a `ctypes` stub — see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain pointers and sizes; every pointer is a DEVICE pointer into caller-owned, contiguous memory
 *     (16-byte aligned); nothing is allocated or freed inside; work is enqueued asynchronously on `stream`
 *     (a `hipStream_t`, passed as void*; NULL = the default stream); no host synchronisation.
 *   - return value: 0 = VKN_OK, < 0 = error (vkn_strerror); never throws, never aborts.
 *   - re-entrant and thread-safe; scratch memory is the caller's `ws` buffer (size from vkn_stage_workspace_bytes /
 *     vkn_head_workspace_bytes).  Library-owned state, all of it idempotent and per (host thread, device) or per device:
 *       (1) one non-blocking side stream + two events per (host thread, device), created on first use by vkn_head_forward_* /
 *           vkn_head_forward_link_f32 when a tracking link is requested WITHOUT VKN_FLAG_SERIAL_LINK, never destroyed (process
 *           lifetime) and shared by every caller stream of that thread (calls of one thread are host-ordered, so their side-stream
 *           work is ordered too; it does not overlap ACROSS caller streams of one thread).  VKN_FLAG_SERIAL_LINK: nothing is created;
 *       (2) a per-kernel "dynamic LDS limit raised on device d" bit mask (hipFuncSetAttribute once per process and device).
 *     Nothing else is kept between calls; two threads may call concurrently with different workspaces.
 *   - layouts: x [B][C][P] fp32 (NCHW, P = H*W); mask logits [B][N][P] fp32; kernels / object features [B][N][C] fp32
 *     (the reference's [B,N,C,1,1] with conv_kernel_size K = 1, the only value in any shipped config).
 *   - arithmetic: fp32 storage; gather / decode contract on MFMA with an f16 hi+lo operand split and fp32 accumulation
 *     (2^-22 relative operand error, requires |x| < 65504: watched by the workspace status word, VKN_STATUS_RANGE); the [N x C] GEMMs are exact-fp32 MFMA, or bf16 MFMA on a
 *     three-term operand split (2^-24 relative, full fp32 range) when pre-split weights are supplied (VknStageWeights.prepared).
 *     flags & VKN_FLAG_REF_KERNELS selects plain fp32 FMA kernels for gather / decode (slow, exact; debugging).
 */
#ifndef VKN_H
#define VKN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKN_VERSION 0x000600 /* 0.6.0: the training tail on the LOW-RES logits (vkn_assign_costs_lowres_batch_f32, vkn_mask_losses_fwd_lowres_f32 /
                              * _bwd_lowres_f32), vkn_sgd_momentum_f32, one-pass kernel initialisation (VKN_FLAG_INIT_SEPARATE opts out),
                              * vkn_track_link_flags_f32.  0.5.0: the persistent chain runs on the two-term fp16 split (VKN_FLAG_CHAIN_BF16X3 opts out; vkn_prepared_bytes grows by the
                              * fp16 weight images), the loss tail without target tensors (vkn_stage_targets ...), the backward glue entry points,
                              * vkn_sum_n_f32.  0.4.0: few-row chain (VKN_FLAG_CHAIN_KSPLIT), VKN_FLAG_SCALED_F16 + vkn_upsample_bilinear_f16out,
                              * VKN_FLAG_JOIN_EARLY, struct size probes */

#define VKN_OK 0
#define VKN_E_ARG (-1)       /* null pointer / non-positive size */
#define VKN_E_SHAPE (-2)     /* shape outside the supported envelope (see vkn_strerror text) */
#define VKN_E_WORKSPACE (-3) /* ws too small or NULL */
#define VKN_E_LAUNCH (-4)    /* HIP launch error */
#define VKN_E_ALIGN (-5)     /* pointer not 16-byte aligned */
#define VKN_E_RANGE (-6)     /* vkn_workspace_status: the feature map left the f16-split envelope (|x| >= 65504 or non-finite) */

/* Status word = the first four bytes of every stage / head / chain workspace (`ws`).  Kernels only ever OR bits into it; the CALLER
 * zeroes it once (vkn_workspace_init, or a memset of the first 256 bytes) and reads it when it wants to pay for a synchronisation
 * (vkn_workspace_status).  Bits: */
#define VKN_STATUS_RANGE 1u  /* a mask gather of a vkn_stage_* / vkn_head_* call produced a non-finite sum: some |x| >= 65504 (the f16
                              * hi/lo split turns it into inf, and inf x {0,1} reaches EVERY kernel row of that frame) or x itself was
                              * non-finite.  The outputs of that call are garbage.  Checked in the fixed-order reduction that ends
                              * every gather (one compare per [B][N][C] output value: free); the stand-alone op entry points
                              * (vkn_mask_gather_f32, vkn_mask_decode_f32 ...) do not check.
                              * Also set by the persistent [N x C] chain on the two-term fp16 split (its default form) when one of the
                              * activation images it stores UNSCALED — LayerNorm outputs, ReLU / FFN hidden rows, the cls / mask branch
                              * inputs: O(1) for ordinary weights — holds |v| >= 2^15 or a NaN (a large LayerNorm gain, a fine-tuned W1):
                              * re-run with VKN_FLAG_CHAIN_BF16X3 (fp32 range). */

#define VKN_FLAG_REF_KERNELS 1u /* exact-fp32 FMA gather/decode kernels instead of the MFMA ones */
#define VKN_FLAG_EXACT_GEMM 2u  /* exact-fp32 MFMA for the [N x C] GEMMs even when pre-split weights are supplied */
#define VKN_FLAG_LOGITS_HANDOFF 4u /* vkn_head_forward_f32: keep fp32 logits between stages instead of bit words (A/B; same results) */
#define VKN_FLAG_BITS_HANDOFF 16u  /* vkn_head_forward_f32: stage hand-off as bit words through two kernels (decode-bits, gather-bits)
                                      instead of the fused decode -> gather pass over x (A/B; same results) */
/* Storage type of the feature map x.  The head computes in fp32 whatever the storage; with a 2-byte x the x-streaming kernels
 * (gather, fused decode->gather, decode) read half the bytes and drop every MFMA against the (zero) low half of x.  fp16 enters
 * the f16 MFMAs as is; bf16 is converted to f16 on the fly (exact for 2^-14 <= |x| < 65504).  On x' = float(half(x)) the fp32
 * path returns the SAME BITS as the half path on half(x) (tests/test_gpu_xhalf.py), i.e. the only deviation from the fp32
 * reference is the rounding of x itself: |dx / x| <= 2^-11 (fp16) / 2^-8 (bf16) — stated tolerance on mask logits 5e-3 / 4e-2 of
 * the logit scale (tests), not the 1e-3 of the fp32 path.  The reference has no reduced-precision mode.
 * Needs H*W % 64 == 0; not available with VKN_FLAG_REF_KERNELS.  With the flags below `x` points at 2-byte elements. */
#define VKN_X_F32 0
#define VKN_X_F16 1
#define VKN_X_BF16 2
#define VKN_FLAG_X_F16 64u    /* x is [B][C][H*W] fp16 */
#define VKN_FLAG_X_BF16 128u  /* x is [B][C][H*W] bf16 */
/* The [N x C] chain of a stage runs either as one launch per GEMM (k_gemm_s3 ...: the tile stream of every GEMM spread over the chip,
 * best for few rows) or as two persistent row-owner kernels around the attention (k_chain_a / k_chain_c, vkn_chain.hip: C == 256,
 * num_cls_fcs == num_mask_fcs == 1, ff % 256 == 0, pre-split composite weights; best from ~2048 rows = 17 frames of 117 kernels on).
 * Default: by row count.  Same arithmetic (bf16x3 MFMA, fp32 two-pass LayerNorm), different summation order: the two agree to fp32
 * rounding (tests/test_gpu_parity.py::test_persistent_chain_equals_launch_per_gemm_chain), each is deterministic. */
#define VKN_FLAG_CHAIN_LAUNCHES 256u   /* always one launch per GEMM */
#define VKN_FLAG_CHAIN_PERSISTENT 512u /* always the persistent kernels (where the shape allows them) */
#define VKN_FLAG_JOIN_EARLY 32768u     /* vkn_head_forward_*: join the side-stream link BEFORE the x4 upsample instead of behind it (1-3 % slower in back-to-back throughput;
                                        * the tracking embeddings of a single call are complete as early as its masks) */
#define VKN_FLAG_SCALED_F16 16384u     /* vkn_head_forward_*: `scaled_out` is fp16 [B][N][H*S][W*S] (see vkn_upsample_bilinear_f16out); S in {2, 4} */
/* The persistent kernels run on the TWO-term fp16 split of both operands (hi + lo, 3 cross products, 4 bytes per weight: vkn_chain_h2.hip;
 * round 5) wherever vkn_prepare_stage_f32 built the fp16 weight images (the C == 256 shapes): 2^-22 per product instead of the 2^-24 of
 * the three-term bf16 split (6 products, 6 bytes) — against fp64 the chain's outputs sit where torch's own fp32 GEMMs sit
 * (profiles/r05_chain_two_term_accuracy.txt); same parity bounds (tests/test_gpu_parity.py), 29 % less time per stage
 * (profiles/r05_chain_two_term_built.txt).  The one-launch-per-GEMM and few-row forms keep the bf16 split. */
#define VKN_FLAG_CHAIN_BF16X3 65536u   /* the persistent kernels on the three-term bf16 split, as rounds 2-4 ran them (A/B; fp32 exponent range in every operand) */
#define VKN_FLAG_CHAIN_KSPLIT 8192u    /* always the few-row chain: one column-spread launch per GEMM phase, normalisation in the consumer (vkn_ksplit.hip) */
#define VKN_FLAG_SERIAL_LINK 32u   /* vkn_head_forward_f32: run the tracking link on the caller's stream instead of the library's side
                                    * stream (A/B, or callers that must see ONE stream; same results) */
#define VKN_FLAG_CLIP_LINK 8u      /* vkn_head_forward_f32: the B frames are CONSECUTIVE frames of one video: prev_obj is [1][N][C] (the
                                      kernels of the frame before frame 0) and frame b > 0 links to this call's own frame b - 1 */

/* vkn_head_forward_link_f32 with a previous_link block and VKN_FLAG_CLIP_LINK, in phases — for a clip whose frames are sharded over
 * ranks: rank r runs A at once, B when rank r-1's last kernels have arrived (they are `prev_obj`), sends its own last
 * kernels (obj_out[B-1]) on, then runs C.  Same arguments, same workspace and same output tensors in all three calls, nothing else
 * through that workspace in between; no phase bit = the whole call.  A: stages 0 .. S-2 + the last stage's gather; B: the last stage's
 * frame-sequential [N x C] chains (writes obj_out / cls_prob); C: the last decode, the upsample, the tracking link. */
#define VKN_FLAG_INIT_SEPARATE 131072u /* vkn_kernel_init_f32: the round-5 form (two decode launches, a copy, an add pass, a logits gather) instead of the
                                        * one-pass kernel (A/B, tests: bit-identical outputs) */
#define VKN_FLAG_PHASE_A 1024u
#define VKN_FLAG_PHASE_B 2048u
#define VKN_FLAG_PHASE_C 4096u
#define VKN_FLAG_PHASE_MASK (VKN_FLAG_PHASE_A | VKN_FLAG_PHASE_B | VKN_FLAG_PHASE_C)

#define VKN_MAX_FCS 4

/* Problem dimensions shared by the stage / head entry points. */
typedef struct VknDims {
    int B;      /* frames in this call */
    int N;      /* kernels per frame (num_proposals + num_stuff_classes at inference, knet/det/kernel_head.py:256-263) */
    int C;      /* in_channels == feat_channels == out_channels (256 in every shipped config); C % 32 == 0, C <= 256 */
    int H, W;   /* feature-map size (stride 8 of the frame); P = H*W */
    int heads;  /* num_heads of `attention` (8) */
    int ff;     /* feedforward_channels (2048); ff % 32 == 0 */
    int ncls;   /* fc_cls outputs (num_classes with sigmoid focal loss, knet/det/kernel_update_head.py:136-139) */
    int n_cls_fcs, n_mask_fcs; /* num_cls_fcs / num_mask_fcs (1 / 1 in shipped configs), <= VKN_MAX_FCS */
    float thr_logit; /* smallest fp32 z with sigmoid(z) > hard_mask_thr (8.940697e-08 for 0.5); bit = (z >= thr_logit) */
    float ln_eps;    /* LayerNorm eps (1e-5) */
} VknDims;

/* One stage's parameters, torch layouts (Linear weight = [out][in]), fp32 device pointers.  Names follow the reference
 * state-dict keys `mask_head.{s}.<...>`.  `ft_wT` is the only derived tensor: the transpose of
 * feat_transform.conv.weight[:, :, 0, 0], prepared once by the host (weight folding). */
typedef struct VknStageWeights {
    const float *ft_w, *ft_b, *ft_wT;                 /* feat_transform.conv.{weight,bias}; all NULL if feat_transform is None */
    const float *dyn_w, *dyn_b;                       /* kernel_update_conv.dynamic_layer   [2C][C] */
    const float *inp_w, *inp_b;                       /* kernel_update_conv.input_layer     [2C][C] */
    const float *ig_w, *ig_b, *ug_w, *ug_b;           /* kernel_update_conv.{input_gate,update_gate} [C][C] */
    const float *norm_in_w, *norm_in_b;               /* kernel_update_conv.norm_in        (LN of update_gate) */
    const float *norm_out_w, *norm_out_b;             /* kernel_update_conv.norm_out       (LN of param_out) */
    const float *inorm_in_w, *inorm_in_b;             /* kernel_update_conv.input_norm_in  (LN of input_gate) */
    const float *inorm_out_w, *inorm_out_b;           /* kernel_update_conv.input_norm_out (LN of input_out) */
    const float *fc_w, *fc_b, *fc_norm_w, *fc_norm_b; /* kernel_update_conv.{fc_layer,fc_norm} */
    const float *attn_in_w, *attn_in_b;               /* attention.attn.in_proj_{weight,bias} [3C][C] */
    const float *attn_out_w, *attn_out_b;             /* attention.attn.out_proj */
    const float *attn_norm_w, *attn_norm_b;           /* attention_norm */
    const float *ffn1_w, *ffn1_b;                     /* ffn.layers.0.0 [ff][C] */
    const float *ffn2_w, *ffn2_b;                     /* ffn.layers.1   [C][ff] */
    const float *ffn_norm_w, *ffn_norm_b;             /* ffn_norm */
    const float *cls_fc_w[VKN_MAX_FCS], *cls_ln_w[VKN_MAX_FCS], *cls_ln_b[VKN_MAX_FCS];    /* cls_fcs.{3i},{3i+1} */
    const float *fc_cls_w, *fc_cls_b;                 /* fc_cls [ncls][C] */
    const float *mask_fc_w[VKN_MAX_FCS], *mask_ln_w[VKN_MAX_FCS], *mask_ln_b[VKN_MAX_FCS]; /* mask_fcs.{3i},{3i+1} */
    const float *fc_mask_w, *fc_mask_b;               /* fc_mask [C][C] */
    /* video tracking link, previous_type == "ffn" (knet/video/kernel_update_head.py:173-190); all NULL for the image head */
    const float *pa_in_w, *pa_in_b, *pa_out_w, *pa_out_b, *pa_norm_w, *pa_norm_b; /* attention_previous(.attn), _norm */
    const float *lffn1_w, *lffn1_b, *lffn2_w, *lffn2_b, *lffn_norm_w, *lffn_norm_b; /* link_ffn, link_ffn_norm */
    /* Optional: every Linear weight above pre-split into three bf16 terms (vkn_prepare_stage_f32 fills a caller-owned buffer of
     * vkn_prepared_bytes).  When set (and VKN_FLAG_EXACT_GEMM is not), the [N x C] GEMMs run on bf16 MFMA with six cross
     * products (2^-24 relative, full fp32 range) instead of exact-fp32 MFMA: ~2.7x faster, fp32-class accuracy.
     * Must be regenerated whenever a weight changes. */
    const void* prepared;
    size_t prepared_bytes;
} VknStageWeights;

int vkn_version(void);
const char* vkn_strerror(int code);
/* zero the 256-byte header of a workspace (asynchronous on `stream`) / synchronise `stream`, read AND CLEAR the status word:
 * VKN_OK, or VKN_E_RANGE when VKN_STATUS_RANGE was set since the last clear.
 * The header exists in the workspaces of the STAGE-SHAPED entry points only (vkn_stage_*, vkn_head_*, vkn_link_block_f32,
 * vkn_track_link_f32, vkn_kernel_updator_f32, vkn_query_merge_f32: they reserve the first 256 bytes and only ever OR into the first
 * word).  Every other entry point with a `ws` argument (gather / decode / kernel-init / panoptic / merge / assignment / linear) uses its
 * workspace from offset 0: a caller that shares ONE buffer between the two kinds hands those `ws + 256` (what the Python binding
 * does), or the status word reads their scratch data. */
int vkn_workspace_init(void* ws, size_t ws_bytes, void* stream);
int vkn_workspace_status(void* ws, size_t ws_bytes, void* stream);
/* sizeof(VknDims) / sizeof(VknStageWeights) as compiled — lets a foreign-language binding verify its struct mirror */
size_t vkn_sizeof_dims(void);
size_t vkn_sizeof_stage_weights(void);

/* ---- op (i): mask gather.  Replaces `sigmoid_masks = (mask_preds.sigmoid() > hard_mask_thr).float();
 *      x_feat = torch.einsum('bnhw,bchw->bnc', sigmoid_masks, x)`  knet/det/kernel_update_head.py:190-195
 *      (same op at knet/video/kernel_iter_head.py:566-571, knet/det/kernel_head.py:248).
 *      xraw_out [B][N][C], cnt_out [B][N] (number of ON pixels, may be NULL).  ws: vkn_gather_workspace_bytes. */
size_t vkn_gather_workspace_bytes(int B, int N, int C, int P);
int vkn_mask_gather_f32(const float* x, const float* mask_logits, float thr_logit, float* xraw_out, float* cnt_out, int B,
                        int N, int C, int P, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- op (i) with a REAL-valued left operand:  out[b][n][c] = sum_p a[b][n][p] x[b][c][p],  asum[b][n] = sum_p a[b][n][p].
 *      Same kernel as vkn_mask_gather_f32 with the mask operand split hi/lo in f16 like x (k_gather_mfma<., 2>).  Three users:
 *      (1) the BACKWARD of the mask decode w.r.t. the kernels, dK = dM . x^T (SURVEY.md §8(a) footnote; the autograd of
 *      `F.conv2d(mask_x, mask_feat)` knet/det/kernel_update_head.py:247-260), asum = the bias gradient; (2) soft ground-truth
 *      masks in the assignment costs (knet/det/mask_hungarian_assigner.py:44-54,100-108); (3) `use_binary=False` gather weights
 *      in the kernel-initialisation pass (knet/det/kernel_head.py:243-250).  Requires |a|, |x| < 65504; absolute resolution of a is
 *      6e-8 (f16 subnormal): callers with tiny operands (gradients) scale by a power of two first.  ws: vkn_gather_workspace_bytes. */
int vkn_mask_gather_real_f32(const float* x, const float* a, float* out, float* asum_out, int B, int N, int C, int P, void* ws,
                             size_t ws_bytes, void* stream);

/* ---- op (iii): mask decode.  Replaces the per-image loop `F.conv2d(mask_x[i:i+1], mask_feat[i], padding=K//2)`, K = 1
 *      knet/det/kernel_update_head.py:247-260.   out[b][n][p] = sum_c kernels[b][n][c] x[b][c][p] + bias[b][n]
 *      kernels [B][N][C] fp32, bias [B][N] or NULL, out [B][N][P].  ws: vkn_decode_workspace_bytes (f16 planes). */
size_t vkn_decode_workspace_bytes(int B, int N, int C);
int vkn_mask_decode_f32(const float* x, const float* kernels, const float* bias, float* out, int B, int N, int C, int P,
                        void* ws, size_t ws_bytes, unsigned flags, void* stream);
/*      ... with every output multiplied by a DEVICE scalar before it is stored: out = *out_scale * (kernels x + bias).  The backward passes
 *      of a training step scale their gradient operand by a power of two into the f16 split's range; this undoes it inside the
 *      kernel instead of in a second pass over the [B][N][P] result.  MFMA kernel only (even P, no VKN_FLAG_REF_KERNELS). */
int vkn_mask_decode_scaled_f32(const float* x, const float* kernels, const float* bias, const float* out_scale, float* out, int B,
                               int N, int C, int P, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- the same decode on PRE-SPLIT kernels: kf_hi / kf_lo are f16 planes [B][roundup(N,32)][C] with hi + lo ~= K (rows >= N
 *      are ignored), exactly what the update kernels hand to the decode inside a stage.  Launches the MFMA kernel only
 *      (bench.py times it with HIP events for the roofline figure).  vkn_split_planes_f32 produces the planes. */
int vkn_split_planes_f32(const float* kernels, void* kf_hi, void* kf_lo, int B, int N, int C, void* stream);
int vkn_mask_decode_planes_f32(const float* x, const void* kf_hi, const void* kf_lo, const float* bias, float* out, int B,
                               int N, int C, int P, void* stream);
/*      ... with x stored as x_dtype (VKN_X_F32 / VKN_X_F16 / VKN_X_BF16; see the note at VKN_FLAG_X_F16) */
int vkn_mask_decode_planes_x(const void* x, int x_dtype, const void* kf_hi, const void* kf_lo, const float* bias, float* out, int B,
                             int N, int C, int P, void* stream);

/* ---- ops (iii) of stage s and (i) of stage s + 1 as ONE pass over x (k_fused_dg, csrc/vkn_fused.hip): the decode
 *      `F.conv2d(mask_x[i:i+1], mask_feat[i])` knet/det/kernel_update_head.py:247-260, the next stage's binarisation
 *      `sigmoid(mask_preds) > hard_mask_thr` :190-192 and its gather `einsum('bnhw,bchw->bnc')` :195.
 *          xraw[b][n][c] = sum_p [ bias[b][n] + sum_c' K[b][n][c'] x[b][c'][p]  >=  thr_logit ] * x[b][c][p],  cnt = the ON count
 *      kf_hi / kf_lo: pre-split kernels as for vkn_mask_decode_planes_f32.  Results are bit-identical to
 *      vkn_mask_decode_planes_f32 followed by vkn_mask_gather_f32.  Needs P % 64 == 0 and C in {64, 128, 256}
 *      (vkn_decode_gather_supported), else VKN_E_SHAPE.  ws: vkn_gather_workspace_bytes. */
int vkn_decode_gather_supported(int C, int P);
int vkn_decode_gather_f32(const float* x, const void* kf_hi, const void* kf_lo, const float* bias, float thr_logit,
                          float* xraw_out, float* cnt_out, int B, int N, int C, int P, void* ws, size_t ws_bytes, void* stream);
int vkn_decode_gather_x(const void* x, int x_dtype, const void* kf_hi, const void* kf_lo, const float* bias, float thr_logit,
                        float* xraw_out, float* cnt_out, int B, int N, int C, int P, void* ws, size_t ws_bytes, void* stream);

/* ---- `F.interpolate(mask_preds, scale_factor=S, mode='bilinear', align_corners=False)`
 *      knet/det/kernel_iter_head.py:122-130.  in [planes][H][W] -> out [planes][H*S][W*S]. */
int vkn_upsample_bilinear_f32(const float* in, float* out, int planes, int H, int W, int S, void* stream);
/*      ... with the result stored as fp16 (opt-in): the same fp32 interpolation, ONE round-to-nearest at the store; S in {2, 4},
 *      W * S % 4 == 0, out 8-byte aligned.  Halves the bytes of the largest write of a head step (245 MB per 1024x2048 frame at x4).
 *      |error| <= 2^-11 |logit| against vkn_upsample_bilinear_f32 (bit-identical to its result rounded to fp16); the sign — the
 *      binary mask — is preserved for every |logit| >= 6e-8. */
int vkn_upsample_bilinear_f16out(const float* in, void* out_f16, int planes, int H, int W, int S, void* stream);
/*      its adjoint (training: the losses act on the up-scaled predictions): grad_out [planes][H*S][W*S] -> grad_in [planes][H][W];
 *      S in {1, 2, 3, 4, 8} (VKN_E_SHAPE otherwise) */
int vkn_upsample_bilinear_bwd_f32(const float* grad_out, float* grad_in, int planes, int H, int W, int S, void* stream);

/* ---- sigmoid focal loss of the classification branch (mmdet FocalLoss(use_sigmoid=True), the `loss_cls` of every shipped config:
 *      configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:131-136; call site knet/det/kernel_update_head.py:296-300).
 *      logits [M][ncls], labels int64 [M] (ncls or anything outside [0, ncls) = background); weight: NULL, [M] (per row,
 *      weight_elementwise = 0) or [M][ncls] (weight_elementwise = 1: `label_weights` of KernelUpdateHead.get_targets, :396-441).
 *      partial [vkn_focal_loss_blocks(M, ncls)]: block sums of the weighted element losses (fixed order; the caller adds them and
 *      applies loss_weight / avg_factor); grad [M][ncls] = d(sum of the element losses) / d logits. */
int vkn_focal_loss_blocks(int M, int ncls);
int vkn_focal_loss_f32(const float* logits, const long long* labels, const float* weight, int weight_elementwise, int M, int ncls,
                       float alpha, float gamma, float* partial, float* grad, void* stream);

/* ---- weight preparation for the bf16x3 split-MFMA GEMMs: splits every non-NULL Linear weight of `w` (w->prepared is ignored)
 *      into `prepared` (device buffer, >= vkn_prepared_bytes(d, w) bytes, 256-B aligned).  Afterwards set
 *      w->prepared = prepared, w->prepared_bytes = bytes. */
size_t vkn_prepared_bytes(const VknDims* d, const VknStageWeights* w);
int vkn_prepare_stage_f32(const VknDims* d, const VknStageWeights* w, void* prepared, size_t bytes, void* stream);

/* ---- building block (unit tests, micro-benchmarks): out[M][Nout] = act(A[M][K] . W[Nout][K]^T + bias), act 0 none / 1 relu.
 *      w_split = NULL: exact-fp32 MFMA; else the bf16x3 planes of W produced by vkn_split_weight_f32
 *      (6 * roundup(Nout, 256) * K bytes: LDS tile images, K % 32 == 0).  ksplit > 1 splits K over workgroups (needs ws >= ksplit*M*Nout*4 bytes, Nout <= 256). */
int vkn_split_weight_f32(const float* W, void* w_split, int Nout, int K, void* stream);
int vkn_linear_f32(const float* A, const float* W, const void* w_split, const float* bias, float* out, int M, int K, int Nout,
                   int act, int ksplit, void* ws, size_t ws_bytes, void* stream);

/* ---- BACKWARD building blocks of the [B*N x C] chain (training).  With these the chain of a training step — every nn.Linear,
 *      nn.LayerNorm (+ ReLU / sigmoid, + residual) and the attention core of `KernelUpdator.forward` (knet/kernel_updator.py:56-93) and
 *      `KernelUpdateHead.forward` (knet/det/kernel_update_head.py:198-227; the video links knet/video/kernel_update_head.py:324-476) —
 *      runs on this library's kernels in both directions (host side: video-k-net_amd/chain_train.py); csrc/vkn_train.hip.
 *
 *      vkn_split_weight_t_f32: the bf16x3 tile images of the TRANSPOSE of a stored matrix.  W is [K][Nout] row-major (a torch Linear
 *      weight [out = K][in = Nout]); the images describe Wt [Nout][K], so that `vkn_linear_f32(dY, ., images, ...)` with M rows,
 *      K = out features, Nout = in features computes dA = dY . W.  K % 32 == 0; 6 * roundup(Nout, 256) * K bytes.
 *      vkn_linear_dw_f32: dW[n][k] (+)= sum_m dY[m][n] A[m][k]  ([Nout][K], the weight gradient of Y = A W^T),
 *      db[n] (+)= sum_m dY[m][n] (or NULL); ldy / lda = row strides of dY / A in floats; accumulate != 0 adds to dW / db.  Exact-fp32
 *      MFMA (v_mfma_f32_32x32x2_f32), deterministic. */
int vkn_split_weight_t_f32(const float* W, void* w_split_t, int Nout, int K, void* stream);
/*      ... of MANY matrices in ONE launch (training: the images of both orientations of every Linear weight of a stage are rebuilt every
 *      step).  An item describes the matrix Wm [Nout][K] the images stand for through strides into the stored tensor: element (n, k)
 *      is W[n * ldn + k * ldk] for k < kvalid and 0 for kvalid <= k < K (K % 32 == 0: zero padding of a short contraction).
 *      A Linear weight [Nout][K]: ldn = K, ldk = 1, kvalid = K; its transpose, W stored [R][Cc]: Nout = Cc, K = roundup(R, 32), ldn = 1,
 *      ldk = Cc, kvalid = R.  images: 6 * roundup(Nout, 256) * K bytes each, 16-byte aligned.  `items` is a HOST array. */
#define VKN_SPLIT_MAX_ITEMS 64
typedef struct VknSplitItem {
    const float* W;
    void* images;
    long long ldn, ldk;
    int Nout, K, kvalid, reserved;
} VknSplitItem;
size_t vkn_sizeof_split_item(void);
int vkn_split_weights_batch_f32(const VknSplitItem* items, int nitems, void* stream);
int vkn_linear_dw_f32(const float* dY, int ldy, const float* A, int lda, float* dW, float* db, int M, int K, int Nout, int accumulate,
                      void* stream);
/*      ... MANY weight gradients (same M) in ONE launch, each written (not accumulated) into dW [Nout][K] (row stride K) and db [Nout]
 *      (or NULL).  The weight gradients of a chain are off its backward's critical path: the host side queues them and runs them
 *      together at the end (chain_train.py).  `items` is a HOST array. */
#define VKN_DW_MAX_ITEMS 48
typedef struct VknDwItem {
    const float* dY;
    const float* A;
    float* dW;
    float* db;
    int ldy, lda, Nout, K;
} VknDwItem;
size_t vkn_sizeof_dw_item(void);
int vkn_linear_dw_batch_f32(const VknDwItem* items, int nitems, int M, void* stream);
/*      out = act(LayerNorm_C(in + resid) * gamma + beta) per row (resid / gamma / beta may be NULL), act 0 none / 1 ReLU / 2 sigmoid;
 *      stats [M][2] = (mean, 1 / sqrt(var + eps)) for the backward (may be NULL).  C <= 256.  ld* = row strides in floats. */
int vkn_layernorm_act_fwd_f32(const float* in, int ldi, const float* resid, int ldr, const float* gamma, const float* beta, float eps,
                              int act, float* out, int ldo, float* stats, int M, int C, void* stream);
/*      its backward: dx = d loss / d (in + resid) (the same tensor is the gradient of both), dgamma / dbeta [C] (may be NULL; written,
 *      not accumulated; fixed summation order).  `in`, `resid`, `gamma`, `beta`, `stats`: as in the forward call.  One launch. */
int vkn_layernorm_act_bwd_f32(const float* dy, int lddy, const float* in, int ldi, const float* resid, int ldr, const float* gamma,
                              const float* beta, const float* stats, int act, float* dx, int lddx, float* dgamma, float* dbeta, int M,
                              int C, void* stream);
/*      the element-wise core of `KernelUpdator.forward` between its GEMMs (knet/kernel_updator.py:70-90), both directions.
 *      params / inputs: the packed outputs [M][2C] of dynamic_layer / input_layer (first half *_in, second half *_out);
 *      gate_feats [M][C] = param_in * input_in (:70); gates [M][2C] = [input_gate(gate_feats) | update_gate(gate_feats)] before their norms;
 *      features [M][C] = sigmoid(norm_in(update gate)) norm_out(param_out) + sigmoid(input_norm_in(input gate)) input_norm_out(input_out)
 *      (:74-90; gate_sigmoid=True, gate_norm_act=False — the shipped defaults); stats [M][8] for the backward.  C <= 256, C % 4 == 0.
 *      Backward: vkn_updator_mix_bwd_f32 writes d_gates [M][2C], the SECOND halves of d_params / d_inputs [M][2C] and the eight
 *      LayerNorm parameter gradients (d_norms may be NULL); vkn_updator_gate_product_bwd_f32 writes their FIRST halves from d_gate_feats. */
typedef struct VknUpdatorNorms {
    const float *norm_in_w, *norm_in_b, *norm_out_w, *norm_out_b, *input_norm_in_w, *input_norm_in_b, *input_norm_out_w, *input_norm_out_b;
    const float *input_gate_b, *update_gate_b;   /* the gate layers' biases (or NULL): `gates` is the bias-free GEMM output, they are added here */
} VknUpdatorNorms;
typedef struct VknUpdatorNormGrads {
    float *norm_in_w, *norm_in_b, *norm_out_w, *norm_out_b, *input_norm_in_w, *input_norm_in_b, *input_norm_out_w, *input_norm_out_b;
} VknUpdatorNormGrads;
size_t vkn_sizeof_updator_norms(void);       /* sizeof(VknUpdatorNorms) / sizeof(VknUpdatorNormGrads) as compiled (binding self-check) */
size_t vkn_sizeof_updator_norm_grads(void);
int vkn_updator_gate_product_f32(const float* params, const float* inputs, float* gate_feats, int M, int C, void* stream);
int vkn_updator_gate_product_bwd_f32(const float* d_gate_feats, const float* params, const float* inputs, float* d_params, float* d_inputs,
                                     int M, int C, void* stream);
int vkn_updator_mix_fwd_f32(const float* gates, const float* params, const float* inputs, const VknUpdatorNorms* norms, float eps,
                            float* features, float* stats, int M, int C, void* stream);
int vkn_updator_mix_bwd_f32(const float* d_features, const float* gates, const float* params, const float* inputs,
                            const VknUpdatorNorms* norms, const float* stats, float* d_gates, float* d_params, float* d_inputs,
                            const VknUpdatorNormGrads* d_norms, int M, int C, void* stream);
/*      the attention core of nn.MultiheadAttention: out[b][i][h] = softmax_j(q_i . k_j / sqrt(hd)) v_j per frame b and head h.
 *      Q rows b * Nq + i, K / V rows b * Nk + j; head h = columns [h * hd, (h + 1) * hd) of every operand; ld* = row strides
 *      (q, k, v may be column slices of one packed in_proj output).  hd in {4, 8, 16, 32, 64} for the backward, Nk <= 256.
 *      Backward: dQ, dK, dV from dO and the forward's O (the softmax is recomputed in fp32 from q, k). */
int vkn_attention_f32(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* out, int ldo, int B, int Nq, int Nk,
                      int heads, int hd, void* stream);
int vkn_attention_bwd_f32(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo, const float* dO,
                          int lddo, float* dQ, int lddq, float* dK, float* dV, int lddkv, int B, int Nq, int Nk, int heads, int hd,
                          void* stream);

/* ---- the gated kernel update alone.  Replaces `KernelUpdator.forward(update_feature, input_feature)`
 *      knet/kernel_updator.py:56-93 (gate_sigmoid=True, gate_norm_act=False, activate_out=False — the defaults, :15-17).
 *      update_feature [B][N][C] (= x_feat), input_feature [B][N][C] (= kernels, K*K = 1) -> out [B][N][C].
 *      Only the kernel_update_conv.* members of `w` are read.  ws: vkn_stage_workspace_bytes. */
int vkn_kernel_updator_f32(const VknDims* d, const VknStageWeights* w, const float* update_feature,
                           const float* input_feature, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- one refinement stage.  Replaces `KernelUpdateHead.forward(x, proposal_feat, mask_preds)`
 *      knet/det/kernel_update_head.py:170-277 and `VideoKernelUpdateHead.forward(..., previous_obj_feats=...)`
 *      knet/video/kernel_update_head.py:281-541 (previous_type="ffn", previous_link=None).
 *      in : x [B][C][P], obj_in [B][N][C] (= proposal_feat), masks_in [B][N][P], prev_obj [B][N][C] or NULL
 *      out: cls_logits [B][N][ncls], masks_out [B][N][P], obj_out [B][N][C],
 *           x_feat_out [B][N][C] or NULL (4th output of the video head), track_out [B][N][C] or NULL (5th; needs prev_obj). */
size_t vkn_stage_workspace_bytes(const VknDims* d);
int vkn_stage_forward_f32(const VknDims* d, const VknStageWeights* w, const float* x, const float* obj_in,
                          const float* masks_in, const float* prev_obj, float* cls_logits, float* masks_out, float* obj_out,
                          float* x_feat_out, float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- the video tracking link alone (previous_type == "ffn"):
 *      track = link_ffn_norm(link_ffn(attention_previous_norm(attention_previous(q=cur, k=v=prev, identity=cur))))
 *      knet/video/kernel_update_head.py:394-415.  cur = this frame's final object features [B][N][C], prev = the previous
 *      frame's [B][N][C].  Lets a clip be processed as one batch: all frames' masks first (they do not depend on the
 *      previous frame), then one link call with prev[b] = cur[b-1].  ws: vkn_stage_workspace_bytes. */
int vkn_track_link_f32(const VknDims* d, const VknStageWeights* w, const float* cur_obj, const float* prev_obj,
                       float* track_out, void* ws, size_t ws_bytes, void* stream);
/* the same with the CHAIN-FORM flags of the head call it completes (VKN_FLAG_CHAIN_KSPLIT / _LAUNCHES / _EXACT_GEMM): the library picks the
 * link's arithmetic by row count, so a one-frame re-link after a multi-frame call (a rank's first frame against its neighbour's last
 * kernels: video_k_net_amd/dist.py) would otherwise run another form than the in-call link of the other frames — fp32 rounding apart.
 * Pass the flag of the form the B-frame call took (its row count decides: see INTEGRATION.md "batch size and bits"). */
int vkn_track_link_flags_f32(const VknDims* d, const VknStageWeights* w, const float* cur_obj, const float* prev_obj,
                             float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- a previous-frame LINK BLOCK of the video head's last stage (knet/video/kernel_update_head.py:192-236, 324-476):
 *        kv  = w has kernel_update_conv.* ? KernelUpdator_w(update_feature, prev) : prev
 *        out = link_ffn_norm(link_ffn(attention_previous_norm(attention_previous(q = cur, k = v = kv, identity = cur))))   (8 heads)
 *      `w` is a VknStageWeights holding ONLY the block's weights, in the members of the modules the block is made of:
 *      kernel_update_conv.* (dyn_w .. fc_norm_b; NULL = no updator), attention_previous.* (pa_*), link_ffn.* (lffn*).
 *      One primitive covers every variant the reference has:
 *        previous_type "ffn"        cur = the stage's updated kernels, no updator                     -> tracking embedding (:394-415)
 *        previous_type "update"     ... updator(update_feature = x_feat, prev)                        -> tracking embedding (:417-445)
 *        previous_type "update_obj" ... updator(update_feature = the updated kernels, prev)           -> tracking embedding (:446-476)
 *        previous_link "update_dynamic_cov"  cur = the stage's INCOMING kernels, updator(x_feat, prev) -> replaces them    (:324-348)
 *        previous_link "link_atten"          cur = the stage's incoming kernels, no updator            -> replaces them    (:350-372)
 *      update_feature, cur, prev, out: [B][N][C].  ws: vkn_stage_workspace_bytes. */
int vkn_link_block_f32(const VknDims* d, const VknStageWeights* w, const float* update_feature, const float* cur,
                       const float* prev, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- clip-level QUERY MERGE of the VIS heads, query_merge_method = "attention" | "attention_pos"
 *      (knet_vis/tracker/kernel_frame_iter_head.py:142-160 on the per-frame object features;
 *       knet_vis/tracker/kernel_update_head.py:244-263 on the per-frame gathers):
 *        out = query_merge_ffn_norm(query_merge_ffn(query_merge_norm(
 *                  query_merge_attn(query, key = value = keys, query_pos = pos, key_pos = pos repeated per frame))))      (8 heads)
 *      d: B = clips, N = queries (= kernels) per clip, ff = the merge FFN's width (8 C in the reference).
 *      query [B][N][C]; keys [B][num_frames * N][C] (frame-major: key f * N + n is kernel n of frame f); pos [N][C] or NULL;
 *      out [B][N][C].  `w`: only pa_* (query_merge_attn.attn.*, query_merge_norm.*) and lffn* (query_merge_ffn.*, query_merge_ffn_norm.*).
 *      num_frames * N may exceed 256 (up to 10240 keys).  ws: vkn_query_merge_workspace_bytes. */
size_t vkn_query_merge_workspace_bytes(const VknDims* d, int num_frames);
int vkn_query_merge_f32(const VknDims* d, int num_frames, const VknStageWeights* w, const float* query, const float* keys,
                        const float* pos, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- vkn_stage_forward_f32 for the heads with previous_link / previous_type = "update" | "update_obj"
 *      (`VideoKernelUpdateHead.forward`, knet/video/kernel_update_head.py:281-541, all branches): link_pre (or NULL) rewrites obj_in
 *      from prev_obj before the update, link_track (or NULL = the stage's own "ffn" link) produces track_out; track_src: update
 *      feature of link_track's updator, 1 = x_feat ("update"), 2 = the updated kernels ("update_obj"). */
int vkn_stage_forward_link_f32(const VknDims* d, const VknStageWeights* w, const VknStageWeights* link_pre,
                               const VknStageWeights* link_track, int track_src, const float* x, const float* obj_in,
                               const float* masks_in, const float* prev_obj, float* cls_logits, float* masks_out, float* obj_out,
                               float* x_feat_out, float* track_out, void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- the [B*N, C] chain of one stage ALONE (ops ii-a, ii-b and the cls / mask FC branches of vkn_stage_forward_f32, no gather,
 *      no decode): x_feat [B][N][C] is given (already feat-transformed), the folded fp32 decode kernels Kf = fc_mask(.) . W_ft
 *      [B][N][C] and bias kb = fc_mask(.) . b_ft [B][N] are returned.  For heads whose gather output is post-processed before the
 *      update — the clip-level VIS heads average x_feat over the frames of a clip (`query_merge_method='mean'`,
 *      knet_vis/tracker/kernel_update_head.py:240-243) and decode every frame with the SAME kernels (:318-330).  cls_logits may be
 *      NULL (stages built with with_cls=False have no classification branch, :133-146).  d->H, d->W are ignored. */
int vkn_stage_chain_f32(const VknDims* d, const VknStageWeights* w, const float* x_feat, const float* obj_in, float* cls_logits,
                        float* kernels_out, float* kb_out, float* obj_out, void* ws, size_t ws_bytes, unsigned flags,
                        void* stream);

/* ---- the S-stage loop.  Replaces `KernelIterHead.simple_test_mask_preds` knet/det/kernel_iter_head.py:285-311 and
 *      `VideoKernelIterHead.simple_test_mask_preds_plus_previous` knet/video/kernel_iter_head.py:529-564
 *      (stage loop + last-stage bilinear upsample `_mask_forward` :118-137 + cls sigmoid :307-308).
 *      prev_obj is given to the LAST stage only (video :544-546).
 *      out: obj_out [B][N][C], cls_prob [B][N][ncls] (sigmoid applied), mask_preds_out [B][N][P],
 *           scaled_out [B][N][H*up][W*up] or NULL (skipped), track_out [B][N][C] or NULL.
 *      Stage hand-off when H*W % 64 == 0 (and C in {64, 128, 256}): stage s's mask decode and stage s+1's mask gather run as ONE
 *      pass over x (k_fused_dg) — the next gather reads nothing but bit(logit >= thr), so intermediate logits are never
 *      materialised and every output is bit-identical to the other two paths kept for A/B: VKN_FLAG_BITS_HANDOFF (two kernels,
 *      bit words of 1/32 of the logits' bytes in between) and VKN_FLAG_LOGITS_HANDOFF (fp32 logits in between). */
size_t vkn_head_workspace_bytes(const VknDims* d);
int vkn_head_forward_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const float* x,
                         const float* proposal_feats, const float* mask_preds_in, const float* prev_obj, float* obj_out,
                         float* cls_prob, float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out,
                         void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- vkn_head_forward_f32 for the "update" video heads (configs/det/video_knet_kitti_step/video_knet_s3_swin{b,l}_*_joint_update.py:
 *      previous_link="update_dynamic_cov", previous_type="update"; ..._update_conv_short_track_fc.py: "update_dynamic_cov" + "ffn").
 *      link_pre / link_track / track_src as in vkn_stage_forward_link_f32, applied in the LAST stage (knet/video/kernel_iter_head.py:
 *      544-546).  With link_pre the masks of frame t depend on the FINAL kernels of frame t-1: under VKN_FLAG_CLIP_LINK (B consecutive
 *      frames) the last stage's [N x C] chain runs frame by frame between the batched last gather and the batched last decode;
 *      stages 0..S-2 and every pass over x stay batched.  Without the flag prev_obj is [B][N][C] and the frames are independent. */
int vkn_head_forward_link_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const VknStageWeights* link_pre,
                              const VknStageWeights* link_track, int track_src, const float* x, const float* proposal_feats,
                              const float* mask_preds_in, const float* prev_obj, float* obj_out, float* cls_prob,
                              float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out, void* ws,
                              size_t ws_bytes, unsigned flags, void* stream);

/* ---- the same call with two caller-owned hipEvent_t recorded on `stream` immediately before / after the LAST stage's mask-decode
 *      launch (either may be NULL): lets a benchmark time the dominant kernel live, inside its timed steps, instead of in a
 *      separate loop (bench.py `roofline`). */
int vkn_head_forward_prof_f32(const VknDims* d, int num_stages, const VknStageWeights* stages, const float* x,
                              const float* proposal_feats, const float* mask_preds_in, const float* prev_obj, float* obj_out,
                              float* cls_prob, float* mask_preds_out, float* scaled_out, int upsample_stride, float* track_out,
                              void* ws, size_t ws_bytes, unsigned flags, void* stream, void* ev_decode_start,
                              void* ev_decode_stop);

/* ---- kernel initialisation ("pass 0").  Replaces `ConvKernelHead._decode_init_proposals` AFTER its loc / seg convs
 *      (knet/det/kernel_head.py:204-263; `simple_test_rpn` :506-508), i.e. everything between the localization FPN's two feature
 *      maps and the first `KernelUpdateHead`:
 *        mask_preds[:, :Np]  = init_kernels(loc_feats)                     1x1 conv, no bias, frame-shared kernels   (:222)
 *        seg_preds           = conv_seg(sem_feats)                         1x1 conv + bias                           (:231-234)
 *        x_feats             = sem_feats + loc_feats                                                                (:238-241)
 *        obj_feats           = einsum('bnhw,bchw->bnc', (sigmoid(mask_preds) > 0.5).float(), x_feats)  (use_binary) (:243-250)
 *        proposal_feats      = init_kernels.weight + obj_feats                                                      (:234, :252-254)
 *        cat_stuff (eval):     mask_preds[:, Np:] = seg_preds[:, num_thing_classes:],
 *                              proposal_feats[:, Np:] = conv_seg.weight[num_thing_classes:]                         (:255-263)
 *      in : loc_feats, sem_feats [B][C][P] (sem_feats NULL: semantic_fpn=False, then x_feats = loc_feats);
 *           init_w [Np][C]; seg_w [ncls][C], seg_b [ncls]; with_obj: 0 = proposal_feats_with_obj off, 1 = on with
 *           use_binary=True, 2 = on with use_binary=False (weights (sigmoid(z) > 0.5) * sigmoid(z), :246-247); thr_logit as in VknDims.
 *      out: x_feats [B][C][P]; mask_preds [B][N][P] and proposal_feats [B][N][C] with N = Np + (cat_stuff ? ncls -
 *           num_thing_classes : 0); seg_preds [B][ncls][P] or NULL (kept in the workspace).
 *      flags: VKN_FLAG_X_F16 / VKN_FLAG_X_BF16 — loc_feats, sem_feats AND x_feats are 2-byte elements (the head then reads x_feats with the
 *           same flag): mask_preds / seg_preds are the bits of the fp32 pass on the widened features, x_feats = half(float(sem) +
 *           float(loc)) (one more rounding), proposal_feats the bits of the fp32 gather on that x_feats.  Needs P % 64 == 0, with_obj != 2.
 */
size_t vkn_kernel_init_workspace_bytes(int B, int Np, int ncls, int C, int P);
int vkn_kernel_init_f32(const float* loc_feats, const float* sem_feats, const float* init_w, const float* seg_w,
                        const float* seg_b, int num_thing_classes, int cat_stuff, int with_obj, float thr_logit, float* x_feats,
                        float* mask_preds, float* seg_preds, float* proposal_feats, int B, int Np, int ncls, int C, int P,
                        void* ws, size_t ws_bytes, unsigned flags, void* stream);

/* ---- post-head mask pipeline: joint panoptic merge of one batch of frames, straight from the head's LOW-RES mask logits.
 *      Replaces, per image, `KernelIterHead.get_panoptic` + `merge_stuff_thing_stuff_joint` (knet/det/kernel_iter_head.py:332-370,
 *      467-524; video: knet/video/kernel_iter_head.py:591-640, 832-905) including `KernelUpdateHead.rescale_masks`
 *      (knet/det/kernel_update_head.py:443-458) and the last-stage x`up` interpolate of `_mask_forward` (:122-130): no
 *      [K, ori_h, ori_w] tensor is ever materialised.  `merge_joint=True` semantics (every shipped panoptic config).
 *      All frames of the call share one geometry (img_meta).  Ties between exactly equal scores are broken by the lower index
 *      (torch leaves them unspecified). */
typedef struct VknPanopticCfg {
    int num_proposals;        /* thing kernels = rows [0, num_proposals); stuff kernels = the remaining N - num_proposals rows */
    int num_thing_classes;
    int max_per_img;          /* test_cfg.max_per_img: top-k over (proposal, thing class) pairs */
    float instance_score_thr; /* test_cfg.merge_stuff_thing.instance_score_thr (compared in fp32, as torch does) */
    double overlap_thr;       /* test_cfg.merge_stuff_thing.overlap_thr (compared in fp64, as Python does) */
    int up;                   /* mask_upsample_stride applied to mask_logits first (1: logits are already `scaled_mask_preds`) */
    int Hm, Wm;               /* mask_logits spatial size */
    int Hb, Wb;               /* img_meta['batch_input_shape'] */
    int h, w;                 /* img_meta['img_shape'][:2]   (crop of the batch input) */
    int Ho, Wo;               /* img_meta['ori_shape'][:2]   (output size) */
} VknPanopticCfg;
#define VKN_PANOPTIC_INFO_FIELDS 6
size_t vkn_sizeof_panoptic_cfg(void);
size_t vkn_panoptic_workspace_bytes(const VknPanopticCfg* cfg, int B, int N);
/*      in : cls_prob [B][N][ncls] (sigmoid applied: the head's 2nd output), mask_logits [B][N][Hm][Wm]
 *      out: panoptic_seg int32 [B][Ho][Wo] (0 = void, 1.. = segment ids in creation order);
 *           info int32 [B][K][6], K = max_per_img + (N - num_proposals), one entry per selected kernel k in the reference's
 *           `total_*` order: {mask row, joint label (< num_thing_classes: thing class; else num_thing_classes + stuff index),
 *           segment id or 0, area (#pixels won), original area (#pixels with prob >= 0.5), score bits (fp32)};
 *           nseg int32 [B] = number of segments (-1: internal capacity error, results invalid);
 *           bbox int32 [B][K][4] or NULL: (xmin, ymin, xmax, ymax) of `panoptic_seg == id` for accepted entries, else (-1,-1,10,10)
 *           = `tensor_mask2box` (unitrack/utils/mask.py:80-90) on the segment masks, what the video detector hands its tracker
 *           (knet/video/knet_quansi_dense_embed_fc_joint_train.py:541-584). */
int vkn_panoptic_joint_f32(const VknPanopticCfg* cfg, const float* cls_prob, const float* mask_logits, int B, int N, int ncls,
                           int* panoptic_seg, int* info, int* nseg, int* bbox, void* ws, size_t ws_bytes, void* stream);

/* ---- post-head pipeline, THING-FIRST merge (`merge_joint=False`): `KernelIterHead.merge_stuff_thing`
 *      knet/det/kernel_iter_head.py:385-465 and `VideoKernelIterHead.merge_stuff_thing_thing_first`
 *      knet/video/kernel_iter_head.py:656-742.  Boolean full-resolution masks (1 byte per pixel, [K][HW]) are pasted in score order:
 *      things (order = argsort(-thing_scores); stop at the first score < instance_score_thr; skip empty masks and masks whose
 *      overlap with the painted area exceeds iou_thr of their own area; paint the still-empty part), then stuff (order = the
 *      distinct labels by descending score, one OR-ed mask per label; painted where still empty if that area >= stuff_max_area).
 *      All on the device (2 launches per mask, no host synchronisation); out: panoptic_seg [HW] int32 (0 = void),
 *      info [(Kt + Ks)][5] = per step {segment id or 0, kind 0 thing / 1 stuff, label, instance index (things) or area (stuff),
 *      score bits (things)}, nseg = number of segments.  ws: vkn_merge_workspace_bytes(Kt, Ks). */
size_t vkn_merge_workspace_bytes(int Kt, int Ks);
int vkn_panoptic_thing_first_u8(const unsigned char* thing_masks, const float* thing_scores, const int* thing_labels,
                                const int* thing_order, int Kt, const unsigned char* stuff_masks, const int* stuff_labels,
                                const int* stuff_order, int Ks, int HW, double instance_score_thr, double iou_thr,
                                int stuff_max_area, int* panoptic_seg, int* info, int* nseg, void* ws, size_t ws_bytes,
                                void* stream);

/* ---- train-time one-to-one assignment, one image.  Replaces `MaskHungarianAssigner.assign`
 *      (knet/det/mask_hungarian_assigner.py:160-274; call site knet/det/kernel_iter_head.py:193-207) with the shipped costs
 *      FocalLossCost (mmdet 2.18) + DiceCost (pred_act, :37-74) + MaskCost (pred_act, :87-113):
 *        cost[n][g] = cls_weight * focal(cls_logits)[n][gt_labels[g]] + dice_weight * dice(n, g) + mask_weight * maskcost(n, g)
 *      vkn_assign_costs_f32 computes the cost matrix on the GPU (the two [N x P].[P x G] contractions run on the gather kernel);
 *      vkn_lsap_f32 is a HOST function: scipy.optimize.linear_sum_assignment's algorithm (same scan order and tie rule).
 *      in : mask_logits [N][P] (the kernels' mask predictions at the assign resolution), cls_logits [N][ncls] or NULL,
 *           gt_masks [G][P] with values 0 / 1, gt_labels int32 [G]; N <= 256, G <= 256 (more than 128
 *           predictions: the two activations take one gather launch each instead of sharing one).
 *      out: cost [N][G] fp32 (device).  The caller copies it to the host, runs vkn_lsap_f32 and sets
 *           assigned_gt_inds[row] = col + 1 (0 = background), as the reference does (:262-271). */
typedef struct VknAssignCfg {
    float cls_weight, dice_weight, mask_weight; /* the three `weight=` of train_cfg.assigner */
    float focal_alpha, focal_gamma, focal_eps;  /* FocalLossCost defaults: 0.25, 2, 1e-12 */
    float dice_eps;                             /* DiceCost eps: 1e-3 */
    float dice_pred_min, mask_pred_min;         /* lower clamp of sigmoid(mask logits) inside DiceCost / MaskCost: knet/det/
                                                   mask_hungarian_assigner.py:69,101 clamp at 1e-3 / 1e-2; the knet_vis copies of the
                                                   same classes (knet_vis/det/mask_hungarian_assigner.py:69,100) do not clamp: 0 / 0 */
} VknAssignCfg;
size_t vkn_sizeof_assign_cfg(void);
size_t vkn_assign_workspace_bytes(int N, int G, int P);
int vkn_assign_costs_f32(const VknAssignCfg* cfg, const float* mask_logits, const float* cls_logits, const float* gt_masks,
                         const int* gt_labels, int N, int G, int ncls, int P, float* cost_out, void* ws, size_t ws_bytes,
                         void* stream);
/*      The images of a training batch in ONE call (HOST array of per-image problems, same N / ncls / P, each its own G): the kernels
 *      of image b run behind those of image b - 1 on `stream` and share `ws` (>= vkn_assign_workspace_bytes(N, max G, P)). */
typedef struct VknAssignProblem {
    const float* mask_logits; /* [N][P] */
    const float* cls_logits;  /* [N][ncls] or NULL */
    const float* gt_masks;    /* [G][P] */
    const int* gt_labels;     /* [G] */
    int G;
    float* cost_out;          /* [N][G] */
} VknAssignProblem;
size_t vkn_sizeof_assign_problem(void);
int vkn_assign_costs_batch_f32(const VknAssignCfg* cfg, const VknAssignProblem* probs, int nprob, int N, int ncls, int P, void* ws,
                               size_t ws_bytes, void* stream);
/*      The same cost matrices straight from the LOW-RES logits (round 6): the reference assigns on
 *      `F.interpolate(mask_preds, scale_factor=S, mode='bilinear', align_corners=False)` (knet/det/kernel_update_head.py:122-130 ->
 *      knet/det/kernel_iter_head.py:150-156,225-226); here `mask_logits` of a problem are the stage's [N][h][w] logits BEFORE that
 *      up-scaling and `gt_masks` [G][S h][S w]: interpolation, activation and both contractions run in ONE kernel for the whole batch
 *      (the up-scaled prediction and its activation plane are never written).  S = 2 or 4, S w % 8 == 0, N, G <= 256,
 *      nprob <= 16 — anything else returns VKN_E_SHAPE (the caller up-scales and calls vkn_assign_costs_batch_f32).  Deterministic;
 *      the partial-sum order does not depend on nprob.  Gmax: the largest of the problems' G. */
size_t vkn_assign_lowres_workspace_bytes(int nprob, int N, int Gmax, int h, int w, int S);
int vkn_assign_costs_lowres_batch_f32(const VknAssignCfg* cfg, const VknAssignProblem* probs, int nprob, int N, int ncls, int h, int w,
                                      int S, void* ws, size_t ws_bytes, void* stream);
/*      cost: HOST fp32 [nr][nc]; writes min(nr, nc) (row, col) pairs sorted by row; returns their number or a negative code */
int vkn_lsap_f32(const float* cost, int nr, int nc, int* row_ind, int* col_ind);
/*      The same algorithm ON THE DEVICE, one wavefront per problem, a batch of problems (the images of a training batch) per launch:
 *      the assignment then needs no device -> host copy at all.  cost: DEVICE fp32 [nr][nc] row-major (nr = predictions, nc = ground
 *      truths; nr, nc <= 256); outputs (DEVICE, each may be NULL): gt_inds int64 [nr] = matched col + 1 or 0 — the reference's
 *      `assigned_gt_inds` (:262-271) — and the min(nr, nc) (row, col) pairs sorted by row, exactly scipy's return value.
 *      status (DEVICE int [nprob], may be NULL): 0 ok, 1 the matrix holds NaN / -inf, 2 infeasible.  On failure the outputs are a
 *      VALID dummy assignment (pair k = (row k, column 0), gt_inds = 1 for the first min(nr, nc) rows, 0 beyond): a caller that
 *      reads the status asynchronously can index with them in bounds until it raises.
 *      Results are identical to vkn_lsap_f32 / scipy including tied matrices (same scan order and tie rule, fp64). */
#define VKN_LSAP_MAX_BATCH 64
typedef struct VknLsapProblem {
    const float* cost;
    int nr, nc;
    long long* gt_inds;
    int* row_ind;
    int* col_ind;
} VknLsapProblem;
size_t vkn_sizeof_lsap_problem(void);
int vkn_lsap_batch_f32(const VknLsapProblem* probs, int nprob, int* status, void* stream);

/* ---- the three MASK losses of a training stage, forward and backward (replaces the torch op sequence of
 *      `KernelUpdateHead.loss`, knet/det/kernel_update_head.py:303-322, for the shipped loss objects):
 *        loss_mask  CrossEntropyLoss(use_sigmoid=True)   mean BCE-with-logits over the K positive rows' pixels
 *        loss_dice  DiceLoss(use_sigmoid, activate, eps)  mean over the K rows of 1 - 2 a / (b + eps + c + eps)
 *        loss_rank  CrossEntropyLoss over the kernel axis, target = largest positive row whose mask target covers the pixel (:311-322)
 *      pred, target: [R = B * Ns][P] fp32 (the up-scaled mask logits and the mask targets of `get_targets`), P % 4 == 0;
 *      pos_rows int64 [K] (ascending positive rows), rowk int32 [R] (index among the positives or -1).
 *      fwd out: row_partial [K][vkn_mask_losses_chunks(P)][4] = partial (sum bce, sum p t, sum p^2, sum t^2);
 *               with_rank: lse [B][P], top int32 [B][P] (covering row or -1), rank_partial [B][vkn_mask_losses_blocks(P)]
 *               (fixed-order partial sums: the caller adds them up and applies weights / means — a handful of K-sized ops).
 *      bwd: grad [R][P] = coef[1] [covered] (softmax_n - onehot) + [row positive] (coef[0] (p - t) + (rowcoef[k][0] t + rowcoef[k][1] p) p (1 - p));
 *           coef (device float[2]) = {g_mask w_mask / (K P), g_rank w_rank / (B P)}, rowcoef (device [K][2]) = the dice chain rule
 *           per row {-2 / (b + c), 4 a / (b + c)^2} x g_dice w_dice / K.  One pass writes the gradient of all three losses. */
int vkn_mask_losses_chunks(int P);
int vkn_mask_losses_blocks(int P);
int vkn_mask_losses_fwd_f32(const float* pred, const float* target, const long long* pos_rows, const int* rowk, int K, int B, int Ns,
                            int P, int with_rank, float* row_partial, float* lse, int* top, float* rank_partial, void* stream);
int vkn_mask_losses_bwd_f32(const float* pred, const float* target, const int* rowk, const float* rowcoef, const float* coef,
                            const float* lse, const int* top, int B, int Ns, int P, int with_rank, float* grad, void* stream);

/* ---- the loss tail of a training stage WITHOUT the target tensors (round 5).  `get_targets` + `loss`
 *      (knet/det/kernel_update_head.py:279-441) build, per stage, labels / label_weights / mask_targets / mask_weights for all
 *      R = B (N + S) rows — a zero-filled [R][H][W] tensor with the matched ground-truth masks scattered into it — and then take the
 *      positive rows out again.  Here the ground truth of the batch stays where it is, in one BANK [G_total][P] (every image's thing
 *      masks and stuff masks, concatenated once per step), and a row's target is named by an index into it:
 *        vkn_stage_targets              one launch: labels int64 [R] (ncls = background), label_weights [R][ncls] (:384-392),
 *                                       row_weight [R] (1 on matched predictions and present stuff rows: `mask_weights` of :381,394
 *                                       as one value per row), rowk int32 [R] (rank among the positives or -1), pos_rows int64 [K]
 *                                       ascending, tgt_row int32 [R] (bank row of the row's mask target or -1).  `imgs` is a HOST
 *                                       array: per image the (row, col) pairs of vkn_lsap_batch_f32 (DEVICE), its labels (DEVICE
 *                                       int64), its present stuff classes (DEVICE int64, values in [T, T + S); anything else sets
 *                                       bit 0 of *status and is treated as class T) and where its masks sit in the bank.
 *                                       S = 0: no stuff targets (weights over all ncls columns); S > 0: thing columns [0, T) only.
 *                                       K = sum_b (k_b + n_sem_b) is known on the host: nothing synchronises.
 *        vkn_mask_losses_fwd_bank_f32   vkn_mask_losses_fwd_f32 with row r's target = bank[tgt_row[r]]
 *        vkn_stage_losses_final_f32     one workgroup: the partial sums of vkn_focal_loss_f32 / the two mask-loss passes -> losses[5] =
 *                                       {loss_cls = w_cls sum / avg_factor, pos_acc (top-1 accuracy of the positive rows in percent,
 *                                       :300-301), loss_mask, loss_dice, loss_rank} and the per-row dice sums a [K], b + c [K] that
 *                                       backward needs.  avg_factor_dev (DEVICE scalar, e.g. an all-reduced count) overrides
 *                                       cfg->avg_factor when not NULL; cls_logits may be NULL (no accuracy).
 *        vkn_mask_losses_bwd_bank_f32   vkn_mask_losses_bwd_f32 with the coefficients formed in the kernel from the upstream
 *                                       gradients g_mask / g_dice / g_rank (DEVICE scalars, NULL = 0) and dice_a / dice_bc
 *        vkn_scale_by_f32               out = in * host_scale * g / d (DEVICE scalars, NULL = 1): backward of sum * weight / avg
 *        vkn_check_range_i64            *status |= flag if any v[i] lies outside [lo, hi) (label validation without a host read) */
#define VKN_TAIL_MAX_IMAGES 64
typedef struct VknTailImage {
    const int* row_ind;         /* [k] matched predictions, ascending */
    const int* col_ind;         /* [k] their ground truths */
    const long long* gt_labels; /* [G] */
    const long long* sem_cls;   /* [n_sem] or NULL */
    int k, n_sem;
    int gt_row0, sem_row0;      /* first bank row of this image's thing / stuff masks */
    int pos0;                   /* positives of the images before this one */
    int reserved;
} VknTailImage;
typedef struct VknTailCfg {
    float w_cls, w_mask, w_dice, dice_eps, w_rank, avg_factor;
    int with_rank;
} VknTailCfg;
size_t vkn_sizeof_tail_image(void);
size_t vkn_sizeof_tail_cfg(void);
int vkn_stage_targets(const VknTailImage* imgs, int B, int N, int S, int T, int ncls, float pos_weight, long long* labels,
                      float* label_weights, float* row_weight, int* rowk, long long* pos_rows, int* tgt_row, int* status, void* stream);
int vkn_mask_losses_fwd_bank_f32(const float* pred, const float* bank, const int* tgt_row, const long long* pos_rows, const int* rowk,
                                 int K, int B, int Ns, int P, int with_rank, float* row_partial, float* lse, int* top,
                                 float* rank_partial, void* stream);
int vkn_stage_losses_final_f32(const VknTailCfg* cfg, const float* avg_factor_dev, const float* focal_partial, int n_focal,
                               const float* row_partial, int K, int nchunk, const float* rank_partial, int n_rank,
                               const float* cls_logits, const long long* labels, const long long* pos_rows, int ncls, int B, int P,
                               float* losses, float* dice_a, float* dice_bc, void* stream);
int vkn_mask_losses_bwd_bank_f32(const float* pred, const float* bank, const int* tgt_row, const int* rowk, const float* dice_a,
                                 const float* dice_bc, const float* g_mask, const float* g_dice, const float* g_rank, float w_mask,
                                 float w_dice, float w_rank, int K, const float* lse, const int* top, int B, int Ns, int P,
                                 int with_rank, float* grad, void* stream);
/* ... and WITHOUT the up-scaled gradient tensor (round 6): the predictions are the LOW-RES logits `low` [B][Ns][h][w]; the losses were taken on
 * their xS bilinear up-scaling (S = 2 or 4; F.interpolate(scale_factor=S, bilinear, align_corners=False), knet/det/kernel_iter_head.py:122-130)
 * against bank masks of [S h][S w]; grad_low [B][Ns][h][w] = d(sum of the weighted losses) / d low — the composition of
 * vkn_mask_losses_bwd_bank_f32 and vkn_upsample_bilinear_bwd_f32 in one pass (the [B Ns][S h][S w] gradient is never written). */
int vkn_mask_losses_bwd_lowres_f32(const float* low, const float* bank, const int* tgt_row, const int* rowk, const float* dice_a,
                                   const float* dice_bc, const float* g_mask, const float* g_dice, const float* g_rank, float w_mask,
                                   float w_dice, float w_rank, int K, const float* lse, const int* top, int B, int Ns, int h, int w, int S,
                                   int with_rank, float* grad_low, void* stream);
/* ... and the FORWARD sums of the same losses straight from the low-res logits: vkn_mask_losses_fwd_bank_f32's outputs for the xS
 * up-scaling of `low` without that tensor (knet/det/kernel_update_head.py:122-130 -> :279-349 loss_mask / loss_dice / loss_rank):
 * row_partial [K][vkn_mask_losses_lowres_chunks(h, w)][4], lse / top [B][S h S w], rank_partial [B][vkn_mask_losses_lowres_chunks(h, w)].
 * S = 2 or 4, Ns <= 256; anything else VKN_E_SHAPE (the caller up-scales and calls vkn_mask_losses_fwd_bank_f32). */
int vkn_mask_losses_lowres_chunks(int h, int w);
int vkn_mask_losses_fwd_lowres_f32(const float* low, const float* bank, const int* tgt_row, const int* rowk, int K, int B, int Ns, int h,
                                   int w, int S, int with_rank, float* row_partial, float* lse, int* top, float* rank_partial,
                                   void* stream);
int vkn_scale_by_f32(const float* in, const float* g, const float* d, float host_scale, float* out, size_t n, void* stream);
/*      The optimizer step of the training row over ONE flat range (a gradient bucket of dist.BucketedGradAllReducer and the parameters
 *      laid out the same way): torch.optim.SGD's rule with momentum (dampening 0, no nesterov) in one pass —
 *          g' = grad * grad_scale + weight_decay * param;  mom = momentum * mom + g';  param -= lr * mom
 *      (the reference trains through mmcv's optimizer hook around torch's optimizers, external/train.py; `grad_scale` = 1 / world
 *      folds the data-parallel mean).  All three pointers 16-byte aligned, DEVICE fp32, n elements. */
int vkn_sgd_momentum_f32(float* param, const float* grad, float* mom, size_t n, float lr, float momentum, float weight_decay,
                         float grad_scale, void* stream);
int vkn_check_range_i64(const long long* v, size_t n, long long lo, long long hi, int flag, int* status, void* stream);

/* ---- glue of the BACKWARD passes of the two x-streaming ops (training; the passes themselves are vkn_mask_decode_scaled_f32 and
 *      vkn_mask_gather_real_f32 with transposed operands — knet/det/kernel_update_head.py:190-195, 247-260 differentiated):
 *        vkn_pow2_scale_f32      scale8[0] = the power of two s with max|t| s in [2^(target_log2 - 1), 2^target_log2) (s = 2^target_log2
 *                                for an all-zero t; exponent clamped to +-100), scale8[4] = 1 / s (DEVICE float[8]: both 16-byte
 *                                aligned).  scratch2: DEVICE unsigned[2], zero before the first use; the kernel re-zeroes it (one
 *                                scratch per stream).  One launch, no host read.
 *        vkn_scale_pad_rows_f32  out [B][Rp][P] = t [B][R][P] * *scale (NULL = 1), rows R .. Rp zero
 *        vkn_transpose_pad_f32   out [B][C][Np] = k [B][N][C]^T * *scale (NULL = 1), columns N .. Np zero
 *        vkn_threshold_rows_f16  rows [B][Np][P] fp16 = (logits [B][N][P] >= thr_logit) ? 1 : 0, rows N .. Np zero
 *        vkn_unscale_rows_f32    dk [B][N][C] = dk_p [B][Np][C] * *scale over the first N rows; dkb [B][N] likewise (may be NULL) */
int vkn_pow2_scale_f32(const float* t, size_t n, int target_log2, float* scale8, unsigned int* scratch2, void* stream);
int vkn_scale_pad_rows_f32(const float* t, const float* scale, int B, int R, int Rp, size_t P, float* out, void* stream);
int vkn_transpose_pad_f32(const float* k, const float* scale, int B, int N, int C, int Np, float* out, void* stream);
int vkn_threshold_rows_f16(const float* logits, float thr_logit, int B, int N, int Np, size_t P, void* rows_f16, void* stream);
int vkn_unscale_rows_f32(const float* dk_p, const float* dkb_p, const float* scale, int B, int N, int Np, int C, float* dk, float* dkb,
                         void* stream);

/*      out [n] = srcs[0] + srcs[1] + ... (HOST array of nsrc <= VKN_SUM_MAX DEVICE pointers), summed in that order, one pass:
 *      the feature map's gradient is the sum of six contributions per training step (three gathers, three decodes) */
#define VKN_SUM_MAX 8
int vkn_sum_n_f32(const float* const* srcs, int nsrc, size_t n, float* out, void* stream);

/* ---- quasi-dense embedding association (the `tracker=dict(type='QuasiDenseEmbedTracker', ...)` of the video configs).  Replaces
 *      `QuasiDenseEmbedTracker.match(bboxes, labels, track_feats, frame_id) -> (bboxes, labels, ids)` together with the `update_memo`
 *      and `memo` it calls: knet/video/qdtrack/trackers/quasi_dense_embed_tracker.py:137-207, :47-103, :105-135 (ctor kwargs :11-38).
 *      One single-workgroup kernel per frame over a device-resident memo (caller-owned `state`, reset once per video): score sort,
 *      duplicate suppression by IoU, [n x m] bi-softmax / softmax / cosine similarity against tracklets + backdrops, the
 *      order-dependent greedy assignment, births, momentum update of the matched tracks, backdrop memo, expiry.
 *      in : bboxes [n][5] (x1, y1, x2, y2, score) fp32, labels [n] int64, embeds [n][embed_dim] fp32 — device pointers, n <= max_dets
 *      out: out_bboxes [max_dets][5], out_labels [max_dets], out_ids [max_dets] — the SURVIVING detections in score order (the
 *           reference returns exactly these rows), ids: >= 0 track id, -1 unmatched (kept as backdrop candidate), -2 suppressed;
 *           out_count[0] = number of surviving detections, out_count[1] = status bits OF THIS CALL (1: tracklet table full, a birth
 *           was dropped — its id is still consumed); the state header keeps the union over all calls since the reset.
 *      Equal scores are ordered by input row (torch's unstable sort leaves that order unspecified).  memo_keep = float(1.0 -
 *      double(memo_momentum)): Python evaluates `1 - self.memo_momentum` in double before it meets the fp32 tensor. */
typedef struct VknTrackerCfg {
    float init_score_thr, obj_score_thr, match_score_thr;
    float memo_momentum, memo_keep;
    float nms_conf_thr, nms_backdrop_iou_thr, nms_class_iou_thr;
    int memo_tracklet_frames, memo_backdrop_frames;
    int with_cats;      /* 0 / 1 */
    int match_metric;   /* 0 bisoftmax, 1 softmax, 2 cosine */
    int max_dets;       /* capacity: detections per frame (<= 256) */
    int max_tracklets;  /* capacity of the tracklet table (max_tracklets + max(memo_backdrop_frames, 1) * max_dets <= 4096) */
    int embed_dim;
} VknTrackerCfg;
size_t vkn_sizeof_tracker_cfg(void);
size_t vkn_qd_tracker_state_bytes(const VknTrackerCfg* cfg);
size_t vkn_qd_tracker_workspace_bytes(const VknTrackerCfg* cfg);
/* byte offsets inside `state` of {header, id, label, last_frame, acc_frame, bbox[5], velocity[5], embed[E], backdrop count[F],
 * backdrop label[F][D], backdrop bbox[F][D][5], backdrop embed[F][D][E]}; header ints: next id, live tracklets, backdrop frames,
 * status, survivors of the last call, calls.  For introspection (`tracker.tracklets`) only. */
int vkn_qd_tracker_state_layout(const VknTrackerCfg* cfg, size_t* offsets12);
int vkn_qd_tracker_reset(const VknTrackerCfg* cfg, void* state, size_t state_bytes, void* stream);
int vkn_qd_tracker_match_f32(const VknTrackerCfg* cfg, void* state, size_t state_bytes, const float* bboxes, const int64_t* labels,
                             const float* embeds, int n, int frame_id, float* out_bboxes, int64_t* out_labels, int64_t* out_ids,
                             int* out_count, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VKN_H */
