// vkn_loss.hip — the three MASK losses of a training stage in two passes over the up-scaled predictions instead of ~60 element-wise /
// reduction launches (knet/det/kernel_update_head.py:303-322 with the shipped loss objects):
//   loss_mask  CrossEntropyLoss(use_sigmoid=True): mean over the K positive rows' pixels of BCE-with-logits(z, t)      (knet/cross_entropy_loss.py:61-101)
//   loss_dice  DiceLoss(use_sigmoid, activate, eps): mean over the K rows of 1 - 2 a / (b + c), a = sum p t, b = sum p^2 + eps,
//              c = sum t^2 + eps, p = sigmoid(z)                                                                        (mmdet 2.18 dice_loss)
//   loss_rank  CrossEntropyLoss over the kernel axis: per pixel, target = the LARGEST positive row whose mask target covers it
//              (:311-322), ignored where none does; mean over ALL B H W pixels of logsumexp_n z - z[target]            (knet/cross_entropy_loss.py:8-43)
// Forward: k_ml_rows (one workgroup per (positive row, pixel chunk): four partial sums) + k_ml_rank_fwd (a thread owns 4 pixels of a
// frame and walks the Ns rows: online logsumexp, the covering row, per-block partial loss; lse and target are kept for backward).
// Backward: ONE kernel writes the gradient of all three losses into one [R][P] tensor (no zero-fill + add of the sparse mask / dice
// gradients into the dense rank gradient).  Fixed-order reductions, no atomics: deterministic.
#include <hip/hip_runtime.h>

#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

namespace {

// (The exponentials of the three kernels are v_exp_f32 (`__expf`, ~2 ulp): the precise expf was a third of their time — 10^8 calls per
// stage — and its last bits do not reach the losses' 1e-4 / the gradients' 2e-3 tolerances.)
constexpr int ML_CHUNK = 8192;   // pixels per workgroup of k_ml_rows

// partial [K][nchunk][4] = (sum bce, sum p t, sum p^2, sum t^2) of row pos_rows[k] over pixels [chunk * ML_CHUNK, ...)
// tgt_row != NULL: `target` is a BANK of ground-truth masks and row r's target is its row tgt_row[r] (the training tail: no [R][P] target
// tensor is ever built); NULL: target is [R][P].
__global__ __launch_bounds__(256) void k_ml_rows(const float* __restrict__ pred, const float* __restrict__ target,
                                                 const int* __restrict__ tgt_row, const long long* __restrict__ pos_rows, int P,
                                                 int nchunk, float* __restrict__ partial) {
    __shared__ float red[4][4];
    const int k = blockIdx.y, ck = blockIdx.x;
    const size_t row = (size_t)pos_rows[k];
    const float* z = pred + row * P;
    const float* t = target + (tgt_row ? (size_t)tgt_row[row] : row) * P;
    const int p_lo = ck * ML_CHUNK, p_hi = min(P, p_lo + ML_CHUNK);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // 16-byte loads, two of each operand in flight per thread (P % 4 == 0 and 16-byte aligned rows: checked by the entry point)
#pragma unroll 2
    for (int p = p_lo + 4 * threadIdx.x; p < p_hi; p += 1024) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + p), tv = *reinterpret_cast<const f32x4*>(t + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float zz = zv[e], tt = tv[e];
            // F.binary_cross_entropy_with_logits: max(z, 0) - z t + log(1 + exp(-|z|))
            // softplus(-|z|) as v_log_f32 of 1 + v_exp_f32 (absolute error < 1e-7 per element — the argument's rounding — on a loss that
            // is a mean of O(1) terms) and the sigmoid through v_rcp_f32: log1pf + an IEEE division were half of this kernel's instructions
            s0 += fmaxf(zz, 0.f) - zz * tt + __logf(1.0f + __expf(-fabsf(zz)));
            const float pp = __builtin_amdgcn_rcpf(1.0f + __expf(-zz));
            s1 += pp * tt;
            s2 += pp * pp;
            s3 += tt * tt;
        }
    }
    s0 = vkn_wave_sum(s0); s1 = vkn_wave_sum(s1); s2 = vkn_wave_sum(s2); s3 = vkn_wave_sum(s3);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; red[3][w] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float* r = red[threadIdx.x];
        partial[((size_t)k * nchunk + ck) * 4 + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
}

// frame b, pixels ML_PX * (blockIdx.x * ML_PXB + tid) .. : walk the Ns rows.  rowk [R]: index of a row among the positives or -1.
// Two pixels per thread (8-byte loads): twice the waves of the 4-pixel version for the serial online-softmax walk over the rows.
// partial [B][vkn_mask_losses_blocks(P)]: one value per workgroup (512 pixels of a frame).
constexpr int ML_PX = 2;
typedef float ml_vec __attribute__((ext_vector_type(ML_PX)));
typedef int ml_ivec __attribute__((ext_vector_type(ML_PX)));
__global__ __launch_bounds__(256) void k_ml_rank_fwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                     const int* __restrict__ tgt_row, const int* __restrict__ rowk, int Ns, int P,
                                                     float* __restrict__ lse,
                                                     int* __restrict__ top, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int p = ML_PX * (blockIdx.x * 256 + threadIdx.x);
    float loss = 0.f;
    if (p < P) {
        float m[ML_PX], s[ML_PX], zt[ML_PX];
        int tp[ML_PX];
#pragma unroll
        for (int e = 0; e < ML_PX; ++e) { m[e] = -INFINITY; s[e] = 0.f; zt[e] = 0.f; tp[e] = -1; }
#pragma unroll 4
        for (int n = 0; n < Ns; ++n) {
            const size_t off = ((size_t)b * Ns + n) * P + p;
            const ml_vec z = *reinterpret_cast<const ml_vec*>(pred + off);
            const bool pos = rowk[b * Ns + n] >= 0;   // uniform
            // (no branch around the second load, so that the loads of several rows can be in flight: a non-positive row re-reads z)
            const size_t toff = (pos && tgt_row) ? (size_t)tgt_row[b * Ns + n] * P + p : off;
            const ml_vec t = *reinterpret_cast<const ml_vec*>((pos ? target : pred) + toff);
#pragma unroll
            for (int e = 0; e < ML_PX; ++e) {
                const float mn = fmaxf(m[e], z[e]);
                s[e] = s[e] * __expf(m[e] - mn) + __expf(z[e] - mn);
                m[e] = mn;
                if (pos && t[e] != 0.f) { tp[e] = n; zt[e] = z[e]; }
            }
        }
        ml_vec l;
        ml_ivec tv;
#pragma unroll
        for (int e = 0; e < ML_PX; ++e) {
            l[e] = m[e] + logf(s[e]);
            tv[e] = tp[e];
            if (tp[e] >= 0) loss += l[e] - zt[e];
        }
        *reinterpret_cast<ml_vec*>(lse + (size_t)b * P + p) = l;
        *reinterpret_cast<ml_ivec*>(top + (size_t)b * P + p) = tv;
    }
    loss = vkn_wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grad[row][p] = coef[1] * [covered] (softmax_n - onehot) + [row positive] (coef[0] (p - t) + (rowcoef[k][0] t + rowcoef[k][1] p) p (1 - p))
// coef (device): [0] = g_mask w_mask / (K P), [1] = g_rank w_rank / (B P);  rowcoef [K][2]: the dice chain rule per row
// The training tail's form (tl.dice_a != NULL): the coefficients are formed HERE from the upstream gradients (device scalars; NULL = 0)
// and the per-row dice sums the forward kept — rowcoef / coef are not read — and the targets come from the bank (tgt_row).
struct MlTail {
    const int* tgt_row;
    const float *dice_a, *dice_bc, *g_mask, *g_dice, *g_rank;
    float c_mask, c_dice, c_rank;   // w_mask / (K P), w_dice / K, w_rank / (B P)
};
__global__ __launch_bounds__(256) void k_ml_bwd(const float* __restrict__ pred, const float* __restrict__ target,
                                                const int* __restrict__ rowk, const float* __restrict__ rowcoef,
                                                const float* __restrict__ coef, const float* __restrict__ lse,
                                                const int* __restrict__ top, int Ns, int P, int with_rank, float* __restrict__ grad,
                                                const MlTail tl) {
    const int b = blockIdx.y;
    const int p = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (p >= P) return;
    const bool tail = tl.dice_a != nullptr;
    const float cm = tail ? (tl.g_mask ? tl.g_mask[0] * tl.c_mask : 0.f) : coef[0];
    const float cr = !with_rank ? 0.f : tail ? (tl.g_rank ? tl.g_rank[0] * tl.c_rank : 0.f) : coef[1];
    const float gd = (tail && tl.g_dice) ? tl.g_dice[0] * tl.c_dice : 0.f;
    f32x4 l = {0.f, 0.f, 0.f, 0.f};
    int4 tp = make_int4(-1, -1, -1, -1);
    if (with_rank) {
        l = *reinterpret_cast<const f32x4*>(lse + (size_t)b * P + p);
        tp = *reinterpret_cast<const int4*>(top + (size_t)b * P + p);
    }
    const int tpv[4] = {tp.x, tp.y, tp.z, tp.w};
    // the rows are independent here: gridDim.z workgroups share a pixel range's rows (more waves in flight per CU than the 8 a
    // [P / 1024, B] grid gives at training sizes)
    const int rpz = (Ns + gridDim.z - 1) / gridDim.z;
    const int n_hi = min(Ns, (int)(blockIdx.z + 1) * rpz);
#pragma unroll 2
    for (int n = blockIdx.z * rpz; n < n_hi; ++n) {
        const size_t off = ((size_t)b * Ns + n) * P + p;
        const f32x4 z = *reinterpret_cast<const f32x4*>(pred + off);
        const int k = rowk[b * Ns + n];   // uniform
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (with_rank) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (tpv[e] >= 0) g[e] = cr * (__expf(z[e] - l[e]) - (tpv[e] == n ? 1.f : 0.f));
        }
        if (k >= 0) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(target + (tl.tgt_row ? (size_t)tl.tgt_row[b * Ns + n] * P + p : off));
            float ca, cb;
            if (tail) {   // the dice chain rule per row: d/dp of 1 - 2 a / (b + c)
                const float a = tl.dice_a[k], bc = tl.dice_bc[k];
                ca = gd * (-2.0f / bc);
                cb = gd * (4.0f * a / (bc * bc));
            } else {
                ca = rowcoef[2 * k];
                cb = rowcoef[2 * k + 1];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pp = 1.0f / (1.0f + __expf(-z[e]));
                g[e] += cm * (pp - t[e]) + (ca * t[e] + cb * pp) * pp * (1.f - pp);
            }
        }
        *reinterpret_cast<f32x4*>(grad + off) = g;
    }
}

// Adjoint of the xS bilinear upsample (F.interpolate(scale_factor=S, bilinear, align_corners=False)): gin[y][x] = sum over the
// output pixels whose two source rows / columns include (y, x) of their weights x gout.  Gather form: a thread owns one input pixel.
// Source row sy(oy) = max((oy + 0.5) / S - 0.5, 0) touches input row y iff sy is in (y - 1, y + 1), i.e. oy in
// [S y - S/2, S y - S/2 + 2 S): 2 S candidate rows (and columns) per input pixel, each tested with its own (y0, y1, ly) — the same
// clamped source-index formula as the forward kernels, so forward and backward agree on every border case.  The 2 S column
// weights are computed once per thread; terms are added in raster order.  Deterministic (torch's kernel scatters with atomics).
template <int S>
__global__ __launch_bounds__(256) void k_upsample_bwd(const float* __restrict__ gout, float* __restrict__ gin, int H, int W) {
    constexpr int R = 2 * S;
    const int plane = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const int OH = H * S, OW = W * S;
    constexpr float rs = 1.0f / (float)S;
    const float* gp = gout + (size_t)plane * OH * OW;
    const int oy0 = S * y - S / 2, ox0 = S * x - S / 2;
    float wxs[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int ox = ox0 + j;
        const float sx = fmaxf(((float)ox + 0.5f) * rs - 0.5f, 0.f);
        const int x0 = min((int)sx, W - 1), x1 = min(x0 + 1, W - 1);
        const float lx = sx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        wxs[j] = (ox >= 0 && ox < OW) ? wx : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int oy = oy0 + i;
        if (oy < 0 || oy >= OH) continue;
        const float sy = fmaxf(((float)oy + 0.5f) * rs - 0.5f, 0.f);
        const int y0 = min((int)sy, H - 1), y1 = min(y0 + 1, H - 1);
        const float ly = sy - (float)y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        const float* rp = gp + (size_t)oy * OW + ox0;
        float row = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (wxs[j] != 0.f) row += wxs[j] * rp[j];
        acc += wy * row;
    }
    gin[(size_t)plane * H * W + idx] = acc;
}

// ---- the mask-loss backward pass WITHOUT the xS gradient tensor (round 6): d(losses) / d(low-res mask logits) in one kernel.
// k_ml_bwd writes the gradient w.r.t. the up-scaled predictions ([B Ns][S h][S w]: 981 MB per stage at the shipped x4 and four 1024x2048
// frames) and the upsample adjoint reads it back; here a thread owns one S x S block of up-scaled pixels SHIFTED by S / 2 (k_ml_fwd_lr's
// blocks: rows S bi + S / 2 .., bi = -1 .. h - 1): it re-forms their logits from the FOUR low-res taps (bi, bi + 1) x (bj, bj + 1) with
// the compile-time weights (a + 0.5) / S, evaluates the gradient of the three losses there (k_ml_bwd's expressions) and folds the
// S x S values back onto the four taps with the separable adjoint.  A low-res pixel (i, j) collects tap (0, 0) of block (i, j),
// (0, 1) of (i, j - 1), (1, 0) of (i - 1, j) and (1, 1) of (i - 1, j - 1): columns through one wave shift, rows through ONE LDS value per
// wave and kernel row (fixed order, deterministic); at the clamped borders a block's two taps of a direction are the same pixel and
// are summed before the exchange.  Workgroup = LR_NW waves (block rows) x 64 lanes (block columns); the first row / column only feed
// their neighbours: 7 x 63 pixels per workgroup.  lse / top / validity of the block are loaded once, the kernel rows n are walked in
// registers with the next row's four taps requested one row ahead, row metadata comes from LDS.
// (The first form of this kernel — unshifted blocks, a 3 x 3 neighbourhood with per-thread three-tap weights, nine loads and a
// 6-shuffle + 2-LDS-value exchange per row — took 397 us per stage at cfg3 size.)
#define LR_NW 8
#define LR_NSPLIT 4
template <int S>
__global__ __launch_bounds__(64 * LR_NW, 2) void k_ml_bwd_lr(const float* __restrict__ low, const float* __restrict__ bank,
                                                          const int* __restrict__ rowk, const float* __restrict__ lse,
                                                          const int* __restrict__ top, int Ns, int h, int w, int with_rank,
                                                          float* __restrict__ grad_low, const MlTail tl) {
    __shared__ float xch[2][LR_NW][64];
    __shared__ int mk[128], mt[128];
    __shared__ float mca[128], mcb[128];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx.z = frame x LR_NSPLIT + part: the kernel rows of a frame are shared by LR_NSPLIT workgroups per pixel tile (independent rows)
    const int b = blockIdx.z / LR_NSPLIT, part = blockIdx.z - b * LR_NSPLIT;
    const int rpp = (Ns + LR_NSPLIT - 1) / LR_NSPLIT, n_lo = part * rpp, n_hi = min(Ns, n_lo + rpp);
    const int bi = (int)blockIdx.y * (LR_NW - 1) - 1 + wv, bj = (int)blockIdx.x * 63 - 1 + lane;   // this thread's block
    const bool blk = bi <= h - 1 && bj <= w - 1;                                       // ... exists
    const bool own = blk && wv >= 1 && lane >= 1;                                      // ... and low-res pixel (bi, bj) is written here
    const int H = S * h, W = S * w;
    const size_t P = (size_t)H * W, lp = (size_t)h * w;
    const float cm = tl.g_mask ? tl.g_mask[0] * tl.c_mask : 0.f;
    const float cr = (with_rank && tl.g_rank) ? tl.g_rank[0] * tl.c_rank : 0.f;
    const float gd = tl.g_dice ? tl.g_dice[0] * tl.c_dice : 0.f;
    const int Y0 = S * bi + S / 2, X0 = S * bj + S / 2;
    unsigned vmask = 0;
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int c = 0; c < S; ++c)
            if (blk && Y0 + a >= 0 && Y0 + a < H && X0 + c >= 0 && X0 + c < W) vmask |= 1u << (a * S + c);
    // clamped low-res taps (byte offsets of a buffer load; the row's offset rides in a scalar register)
    const int r0 = min(max(bi, 0), h - 1), r1 = min(max(bi + 1, 0), h - 1), c0 = min(max(bj, 0), w - 1), c1 = min(max(bj + 1, 0), w - 1);
    const bool rsame = r0 == r1, csame = c0 == c1;   // border ring: both taps of a direction are one pixel
    const int o00 = (r0 * w + c0) * 4, o01 = (r0 * w + c1) * 4, o10 = (r1 * w + c0) * 4, o11 = (r1 * w + c1) * 4;
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(low + (size_t)b * Ns * lp), 0, (int)((size_t)Ns * lp * 4), 0x00020000);
    // the block's pixels in the up-scaled map (clamped into it), in aligned pairs for S = 4 (a pair is inside or outside as a whole)
    size_t prow[S];
#pragma unroll
    for (int a = 0; a < S; ++a) prow[a] = (size_t)min(max(Y0 + a, 0), H - 1) * W;
    constexpr int PW = S == 4 ? 2 : 1, NP = S / PW;
    int pcol[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) pcol[q] = min(max(X0 + PW * q, 0), W - PW);
    // lse / top of the block's pixels (the frame's, shared by every kernel row).  top as one byte per pixel (Ns <= 256); a pixel that no
    // positive row covers (top = -1) — or that lies outside the map — carries lse = +inf (stored: -lse log2(e) = -inf): its softmax term
    // exp(z - lse) is exactly 0, and
    // a byte that equals no row of THIS workgroup's range [n_lo, n_hi), so that its one-hot term is 0 too: no branch per pixel
    float l[S][S];
    unsigned tpk[S];
#pragma unroll
    for (int a = 0; a < S; ++a) {
        tpk[a] = 0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float lv[PW];
            int tv[PW];
            if (with_rank) {
                const size_t o = (size_t)b * P + prow[a] + pcol[q];
                if (PW == 2) {
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 l2 = *reinterpret_cast<const f32x2*>(lse + o);
                    const i32x2 t2 = *reinterpret_cast<const i32x2*>(top + o);
                    lv[0] = l2[0]; lv[PW - 1] = l2[1]; tv[0] = t2[0]; tv[PW - 1] = t2[1];
                } else {
                    lv[0] = lse[o];
                    tv[0] = top[o];
                }
            } else {
#pragma unroll
                for (int e = 0; e < PW; ++e) { lv[e] = 0.f; tv[e] = -1; }
            }
#pragma unroll
            for (int e = 0; e < PW; ++e) {
                const int c = PW * q + e;
                const bool ok = ((vmask >> (a * S + c)) & 1u) && tv[e] >= 0;
                l[a][c] = ok ? lv[e] * -1.4426950408889634f : -INFINITY;   // (-lse log2(e): the softmax term is exp2(z log2(e) + this) — one fma + v_exp_f32)
                tpk[a] |= (unsigned)(ok ? tv[e] : (n_lo == 0 ? n_hi & 255 : 0)) << (8 * c);
            }
        }
    }
    // this workgroup's kernel rows: (k, target row, dice coefficients) once, through LDS — inside the row loop they were three DEPENDENT
    // memory round trips per positive row (rowk -> tgt_row / dice_a / dice_bc -> target pixels)
    for (int q = threadIdx.x; q < n_hi - n_lo; q += 64 * LR_NW) {
        const int k = rowk[b * Ns + n_lo + q];
        mk[q] = k;
        float ca = 0.f, cb = 0.f;
        int t = 0;
        if (k >= 0) {
            const float a_ = tl.dice_a[k], bc = tl.dice_bc[k];
            ca = gd * (-2.0f / bc);
            cb = gd * (4.0f * a_ / (bc * bc));
            t = tl.tgt_row[b * Ns + n_lo + q];
        }
        mt[q] = t; mca[q] = ca; mcb[q] = cb;
    }
    __syncthreads();
    float vn[4];
    if (n_lo < n_hi) {
        const int so = n_lo * (int)lp * 4;
        vn[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, so, 0));
        vn[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, so, 0));
        vn[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, so, 0));
        vn[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, so, 0));
    }
    for (int n = n_lo; n < n_hi; ++n) {
        const float v00 = vn[0], v01 = vn[1], v10 = vn[2], v11 = vn[3];
        if (n + 1 < n_hi) {
            const int so = (n + 1) * (int)lp * 4;
            vn[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, so, 0));
            vn[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, so, 0));
            vn[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, so, 0));
            vn[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, so, 0));
        }
        const int k = __builtin_amdgcn_readfirstlane(mk[n - n_lo]);   // uniform
        float z[S][S], g[S][S];
        {
            constexpr float rs = 1.0f / (float)S;
            float h0[S], h1[S];
#pragma unroll
            for (int c = 0; c < S; ++c) {
                const float lx = ((float)c + 0.5f) * rs;
                h0[c] = (1.f - lx) * v00 + lx * v01;
                h1[c] = (1.f - lx) * v10 + lx * v11;
            }
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float ly = ((float)a + 0.5f) * rs;
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    z[a][c] = (1.f - ly) * h0[c] + ly * h1[c];
                    const int tpv = (int)((tpk[a] >> (8 * c)) & 255u);
                    g[a][c] = cr * (__builtin_amdgcn_exp2f(__builtin_fmaf(z[a][c], 1.4426950408889634f, l[a][c])) - (tpv == n ? 1.f : 0.f));
                }
            }
        }
        if (k >= 0) {   // ONE uniform branch per row: the mask / dice terms of a positive row
            const float ca = mca[n - n_lo], cb = mcb[n - n_lo];
            const float* trow = bank + (size_t)mt[n - n_lo] * P;
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    float tt[PW];
                    if (PW == 2) {
                        const f32x2 t2 = *reinterpret_cast<const f32x2*>(trow + prow[a] + pcol[q]);
                        tt[0] = t2[0]; tt[PW - 1] = t2[1];
                    } else {
                        tt[0] = trow[prow[a] + pcol[q]];
                    }
#pragma unroll
                    for (int e = 0; e < PW; ++e) {
                        const int c = PW * q + e;
                        const float t = tt[e];
                        const float pp = __builtin_amdgcn_rcpf(1.0f + __expf(-z[a][c]));   // (v_rcp_f32: 1 ulp, as k_ml_rows)
                        const float gm = cm * (pp - t) + (ca * t + cb * pp) * pp * (1.f - pp);
                        g[a][c] += ((vmask >> (a * S + c)) & 1u) ? gm : 0.f;
                    }
                }
        }
        // the adjoint onto the four taps, columns first
        float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
        {
            constexpr float rs = 1.0f / (float)S;
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float ly = ((float)a + 0.5f) * rs;
                float q0 = 0.f, q1 = 0.f;
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const float lx = ((float)c + 0.5f) * rs;
                    q0 += (1.f - lx) * g[a][c];
                    q1 += lx * g[a][c];
                }
                p00 += (1.f - ly) * q0;
                p01 += (1.f - ly) * q1;
                p10 += ly * q0;
                p11 += ly * q1;
            }
        }
        // clamped borders: the two taps of a direction are ONE pixel — the tap that the exchange routes to this block's own pixel / row
        // keeps the sum, the other carries nothing (bi = -1: both rows are pixel row 0 = "row bi + 1"; bi = h - 1: both are row h - 1 = "row bi")
        if (rsame) {
            if (bi < 0) { p10 += p00; p11 += p01; p00 = p01 = 0.f; }
            else { p00 += p10; p01 += p11; p10 = p11 = 0.f; }
        }
        if (csame) {
            if (bj < 0) { p01 += p00; p11 += p10; p00 = p10 = 0.f; }
            else { p00 += p01; p10 += p11; p01 = p11 = 0.f; }
        }
        // columns: pixel column bj collects tap (., 0) of its own block and tap (., 1) of the block to its left
        const float hA = p00 + __shfl_up(p01, 1);   // -> pixel row bi
        const float hB = p10 + __shfl_up(p11, 1);   // -> pixel row bi + 1
        float* xb = &xch[(n - n_lo) & 1][0][0];
        xb[wv * 64 + lane] = hB;
        // (LDS-only barrier: __syncthreads() also fences global memory — a wait for the NEXT row's prefetched loads, every row)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (own) grad_low[((size_t)b * Ns + n) * lp + (size_t)bi * w + bj] = hA + xb[(wv - 1) * 64 + lane];
    }
}

// ---- the mask-loss FORWARD pass without the xS tensor (round 6): the row sums of the positive rows (k_ml_rows), the rank loss's
// per-pixel logsumexp / covering row / loss (k_ml_rank_fwd) in ONE kernel that re-forms the up-scaled logits from the low-res ones —
// k_upsample_s's 981 MB write per stage at the shipped x4 and both kernels' reads of it do not exist.
// A thread owns one S x S block of up-scaled pixels SHIFTED by S / 2 (rows S bi + S / 2 .. + S - 1, bi = -1 .. h - 1): exactly the
// pixels whose bilinear source lies between low-res rows bi and bi + 1 / columns bj and bj + 1, with the COMPILE-TIME weights
// (a + 0.5) / S — four low-res loads per kernel row instead of nine, two-tap interpolation (PyTorch's expression
// h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11); at the clamped borders both taps read the same value).  Blocks are numbered flat
// (bi + 1) (w + 1) + bj + 1: no tile of 64 columns is mostly empty.  The kernel rows are walked in registers (online logsumexp with ONE
// exponential per pixel and row), the next row's four values are requested one row ahead, row metadata comes from LDS; a positive
// row's four sums meet per wave (DPP) in LDS and are written once per workgroup: row_partial [K][nchunk][4], nchunk =
// vkn_mask_losses_lowres_chunks(h, w); rank_partial [B][nchunk].
template <int S>
__global__ __launch_bounds__(256, 2) void k_ml_fwd_lr(const float* __restrict__ low, const float* __restrict__ bank,
                                                      const int* __restrict__ tgt_row, const int* __restrict__ rowk, int Ns, int h, int w,
                                                      int with_rank, float* __restrict__ row_partial, int nchunk,
                                                      float* __restrict__ lse, int* __restrict__ top, float* __restrict__ rank_partial) {
    __shared__ int mk[256], mt[256];
    __shared__ float PS[256][4][4];   // [row][wave][sum]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, wg = blockIdx.x;
    const int H = S * h, W = S * w;
    const size_t P = (size_t)H * W, lp = (size_t)h * w;
    for (int q = tid; q < Ns; q += 256) {
        const int k = rowk[b * Ns + q];
        mk[q] = k;
        mt[q] = k >= 0 ? tgt_row[b * Ns + q] : 0;
    }
    const int nblk = (h + 1) * (w + 1);
    const int idx0 = wg * 256 + tid;
    const bool live = idx0 < nblk;
    const int idx = live ? idx0 : nblk - 1;
    const int bi = idx / (w + 1) - 1, bj = idx - (bi + 1) * (w + 1) - 1;
    // pixel (a, c) of the block: row S bi + S / 2 + a, column S bj + S / 2 + c; outside the map for the half blocks of the border ring
    const int Y0 = S * bi + S / 2, X0 = S * bj + S / 2;
    unsigned vmask = 0;
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int c = 0; c < S; ++c)
            if (live && Y0 + a >= 0 && Y0 + a < H && X0 + c >= 0 && X0 + c < W) vmask |= 1u << (a * S + c);
    // clamped low-res taps, as byte offsets of a buffer load (the row's offset rides in a scalar register)
    const int r0 = max(bi, 0), r1 = min(bi + 1, h - 1), c0 = max(bj, 0), c1 = min(bj + 1, w - 1);
    const int o00 = (r0 * w + c0) * 4, o01 = (r0 * w + c1) * 4, o10 = (r1 * w + c0) * 4, o11 = (r1 * w + c1) * 4;
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(low + (size_t)b * Ns * lp), 0, (int)((size_t)Ns * lp * 4), 0x00020000);
    // the block's pixels in the up-scaled map, rows clamped into it; per row the pixels travel in aligned PAIRS (S / 2 is even for
    // S = 4: a pair is inside or outside the map as a whole) or singly (S = 2)
    size_t prow[S];
#pragma unroll
    for (int a = 0; a < S; ++a) prow[a] = (size_t)min(max(Y0 + a, 0), H - 1) * W;
    constexpr int PW = S == 4 ? 2 : 1, NP = S / PW;   // pixels per access, accesses per block row
    int pcol[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) pcol[q] = min(max(X0 + PW * q, 0), W - PW);
    float sm[S][S], zt[S][S];
    unsigned tpk[S];
#pragma unroll
    for (int a = 0; a < S; ++a) {
        tpk[a] = 0;
#pragma unroll
        for (int c = 0; c < S; ++c) { sm[a][c] = 0.f; zt[a][c] = 0.f; }
    }
    unsigned cov = 0;   // bit (a, c): some positive row covers the pixel (its top row is the byte in tpk)
    __syncthreads();
    // the logsumexp's reference point: every logit of the block is a convex combination of its row's four taps, so the maximum of the
    // taps over ALL rows bounds every logit of the block from above — sum exp(z - Mb) cannot overflow and needs no running maximum
    // (ONE fused multiply-add + v_exp_f32 + add per pixel and row instead of the online form's eight operations).  A first walk over
    // the rows (taps only: 4 loads + 4 max per row, eight rows in flight) finds it; a block whose sums underflow all the same (a spread
    // of ~70 between the taps' maximum and the pixel's best row) is redone with the online form at the end.
    float Mb = -INFINITY;
    if (with_rank) {
        for (int n0 = 0; n0 < Ns; n0 += 8) {
            float t[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int so = min(n0 + u, Ns - 1) * (int)lp * 4;
                t[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, so, 0));
                t[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, so, 0));
                t[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, so, 0));
                t[u][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, so, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) Mb = fmaxf(Mb, fmaxf(fmaxf(t[u][0], t[u][1]), fmaxf(t[u][2], t[u][3])));
        }
    }
    constexpr float L2E = 1.4426950408889634f;
    const float mbl = -Mb * L2E;
    float vn[4];
    vn[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, 0, 0));
    vn[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, 0, 0));
    vn[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, 0, 0));
    vn[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, 0, 0));
    for (int n = 0; n < Ns; ++n) {
        const float v00 = vn[0], v01 = vn[1], v10 = vn[2], v11 = vn[3];
        if (n + 1 < Ns) {
            const int so = (n + 1) * (int)lp * 4;
            vn[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, so, 0));
            vn[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, so, 0));
            vn[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, so, 0));
            vn[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, so, 0));
        }
        const int k = __builtin_amdgcn_readfirstlane(mk[n]);   // uniform
        float z[S][S];
        {
            float h0[S], h1[S];
#pragma unroll
            for (int c = 0; c < S; ++c) {
                constexpr float rs = 1.0f / (float)S;
                const float lx = ((float)c + 0.5f) * rs;
                h0[c] = (1.f - lx) * v00 + lx * v01;
                h1[c] = (1.f - lx) * v10 + lx * v11;
            }
#pragma unroll
            for (int a = 0; a < S; ++a) {
                constexpr float rs = 1.0f / (float)S;
                const float ly = ((float)a + 0.5f) * rs;
#pragma unroll
                for (int c = 0; c < S; ++c) z[a][c] = (1.f - ly) * h0[c] + ly * h1[c];
            }
        }
        if (with_rank) {
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int c = 0; c < S; ++c) sm[a][c] += __builtin_amdgcn_exp2f(__builtin_fmaf(z[a][c], L2E, mbl));
        }
        if (k >= 0) {
            const float* trow = bank + (size_t)mt[n] * P;
            float t[S][S];
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (PW == 2) {
                        const f32x2 t2 = *reinterpret_cast<const f32x2*>(trow + prow[a] + pcol[q]);
                        t[a][2 * q] = t2[0];
                        t[a][2 * q + 1] = t2[1];
                    } else {
                        t[a][q] = trow[prow[a] + pcol[q]];
                    }
                }
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const bool ok = (vmask >> (a * S + c)) & 1u;
                    const float zz = z[a][c], tt = ok ? t[a][c] : 0.f;
                    // (k_ml_rows' expressions)
                    const float bce = fmaxf(zz, 0.f) - zz * tt + __logf(1.0f + __expf(-fabsf(zz)));
                    const float pp = __builtin_amdgcn_rcpf(1.0f + __expf(-zz));
                    s0 += ok ? bce : 0.f;
                    s1 += ok ? pp * tt : 0.f;
                    s2 += ok ? pp * pp : 0.f;
                    s3 += tt * tt;
                    if (ok && tt != 0.f) {   // the LAST positive row that covers the pixel (k_ml_rank_fwd)
                        cov |= 1u << (a * S + c);
                        tpk[a] = (tpk[a] & ~(255u << (8 * c))) | ((unsigned)n << (8 * c));
                        zt[a][c] = zz;
                    }
                }
            s0 = vkn_wave_sum(s0); s1 = vkn_wave_sum(s1); s2 = vkn_wave_sum(s2); s3 = vkn_wave_sum(s3);
            if (lane == 0) { PS[n][wv][0] = s0; PS[n][wv][1] = s1; PS[n][wv][2] = s2; PS[n][wv][3] = s3; }
        }
    }
    float loss = 0.f;
    if (with_rank) {
        float m[S][S];
        bool redo = false;
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int c = 0; c < S; ++c) {
                m[a][c] = Mb;
                redo |= !(sm[a][c] > 1e-30f);
            }
        if (redo) {   // (rare) the online form: a running maximum per pixel, one exponential per pixel and row
#pragma unroll
            for (int a = 0; a < S; ++a)
#pragma unroll
                for (int c = 0; c < S; ++c) { m[a][c] = -INFINITY; sm[a][c] = 0.f; }
            for (int n = 0; n < Ns; ++n) {
                const int so = n * (int)lp * 4;
                const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o00, so, 0));
                const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o01, so, 0));
                const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o10, so, 0));
                const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, o11, so, 0));
                constexpr float rs = 1.0f / (float)S;
#pragma unroll
                for (int a = 0; a < S; ++a)
#pragma unroll
                    for (int c = 0; c < S; ++c) {
                        const float lx = ((float)c + 0.5f) * rs, ly = ((float)a + 0.5f) * rs;
                        const float zz = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                        const float d = zz - m[a][c];
                        const float e = __expf(-fabsf(d));
                        sm[a][c] = d > 0.f ? sm[a][c] * e + 1.f : sm[a][c] + e;
                        m[a][c] = fmaxf(m[a][c], zz);
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                float lv[PW];
                int tv[PW];
                bool any = false;
#pragma unroll
                for (int e = 0; e < PW; ++e) {
                    const int c = PW * q + e;
                    const bool ok = (vmask >> (a * S + c)) & 1u, cv = (cov >> (a * S + c)) & 1u;
                    lv[e] = m[a][c] + __logf(sm[a][c]);
                    tv[e] = cv ? (int)((tpk[a] >> (8 * c)) & 255u) : -1;
                    if (ok && cv) loss += lv[e] - zt[a][c];
                    any |= ok;
                }
                if (any) {   // (a pair is inside the map as a whole)
                    const size_t o = (size_t)b * P + prow[a] + pcol[q];
                    if (PW == 2) {
                        *reinterpret_cast<f32x2*>(lse + o) = f32x2{lv[0], lv[PW - 1]};
                        typedef int i32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<i32x2*>(top + o) = i32x2{tv[0], tv[PW - 1]};
                    } else {
                        lse[o] = lv[0];
                        top[o] = tv[0];
                    }
                }
            }
    }
    loss = vkn_wave_sum(loss);
    if (lane == 0) red[wv] = loss;
    __syncthreads();
    if (with_rank && tid == 0) rank_partial[(size_t)b * nchunk + wg] = (red[0] + red[1]) + (red[2] + red[3]);
    for (int q = tid; q < Ns * 4; q += 256) {
        const int n = q >> 2, j = q & 3, k = mk[n];
        if (k >= 0) row_partial[((size_t)k * nchunk + wg) * 4 + j] = (PS[n][0][j] + PS[n][1][j]) + (PS[n][2][j] + PS[n][3][j]);
    }
}

// Sigmoid focal loss (mmdet FocalLoss(use_sigmoid=True): py_sigmoid_focal_loss, the classification loss of every shipped config) over
// logits [M][ncls] with integer labels [M] (label == ncls or out of range = background: an all-zero target row) and optional
// per-row [M] or per-element [M][ncls] weights: element loss = w_row * bce(z, t) * (alpha t + (1 - alpha)(1 - t)) * pt^gamma, pt = (1 - p) t + p (1 - t).
// One pass writes the block partial sums of the loss AND d(sum of losses)/dz per element, so backward is one scaling.
__global__ __launch_bounds__(256) void k_focal(const float* __restrict__ z, const long long* __restrict__ labels,
                                               const float* __restrict__ roww, int elementwise, int M, int ncls, float alpha,
                                               float gamma, float* __restrict__ partial, float* __restrict__ grad) {
    const size_t n = (size_t)M * ncls;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / ncls), c = (int)(i - (size_t)row * ncls);
        const float v = z[i];
        const float t = (labels[row] == (long long)c) ? 1.f : 0.f;
        const float w = roww ? (elementwise ? roww[i] : roww[row]) : 1.f;
        const float e = expf(-fabsf(v));
        const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        const float bce = fmaxf(v, 0.f) - v * t + log1pf(e);
        const float pt = (1.f - p) * t + p * (1.f - t);
        const float at = alpha * t + (1.f - alpha) * (1.f - t);
        const float ptg = powf(pt, gamma);
        acc += w * bce * at * ptg;
        // d/dz: bce' = p - t; pt' = p (1 - p) (1 - 2 t)
        const float dptg = pt > 0.f ? gamma * powf(pt, gamma - 1.f) * p * (1.f - p) * (1.f - 2.f * t) : 0.f;
        grad[i] = w * at * ((p - t) * ptg + bce * dptg);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {   // fixed tree: deterministic
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// The same adjoint for S = 2 and rows that are whole multiples of a wavefront (W % 64 == 0): a thread owns one input column and UB2_R
// consecutive input rows.  Per output row it loads ONE aligned float2 (columns 2x, 2x + 1) and the columns 2x - 1 / 2x + 2 beside it
// (the neighbours' cache lines), forms the horizontal sum once — adjacent input rows share two of their
// four output rows — and blends vertically.  Same weights, same order of additions as k_upsample_bwd<2>: bit-identical, a quarter
// of the load instructions and every output element fetched ~1.25 times instead of 4.
#define UB2_R 4
__global__ __launch_bounds__(256) void k_upsample_bwd2(const float* __restrict__ gout, float* __restrict__ gin, int H, int W) {
    const int plane = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * UB2_R;
    if (x >= W) return;   // (whole waves: W % 64 == 0)
    const int OH = 2 * H, OW = 2 * W;
    const float* gp = gout + (size_t)plane * OH * OW;
    float wxs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = 2 * x - 1 + j;
        const float sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.f);
        const int x0 = min((int)sx, W - 1), x1 = min(x0 + 1, W - 1);
        const float lx = sx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        wxs[j] = (ox >= 0 && ox < OW) ? wx : 0.f;
    }
    // ALL loads of the thread first (ten rows: one aligned float2 + the two outer columns each, from clamped addresses — a use right
    // behind a load, or a branch around one, costs a memory round trip per row: 113 us for the 468 planes of a training stage), then
    // the arithmetic in the order of k_upsample_bwd<2>.  The outer columns hit the cache lines of the neighbours' float2.
    constexpr int NR = 2 * UB2_R + 2;
    float2 vv[NR];
    float vms[NR], vps[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int oy = min(max(2 * y0 - 1 + i, 0), OH - 1);
        const float* rp = gp + (size_t)oy * OW + 2 * x;
        vv[i] = *reinterpret_cast<const float2*>(rp);
        vms[i] = rp[x > 0 ? -1 : 0];
        vps[i] = rp[x + 1 < W ? 2 : 1];
    }
    float hs[NR];   // horizontal sums of output rows 2 y0 - 1 .. 2 y0 + 2 R
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int oy = 2 * y0 - 1 + i;
        float h = 0.f;
        if (oy >= 0 && oy < OH) {   // uniform
            if (wxs[0] != 0.f) h += wxs[0] * vms[i];
            if (wxs[1] != 0.f) h += wxs[1] * vv[i].x;
            if (wxs[2] != 0.f) h += wxs[2] * vv[i].y;
            if (wxs[3] != 0.f) h += wxs[3] * vps[i];
        }
        hs[i] = h;
    }
#pragma unroll
    for (int r = 0; r < UB2_R; ++r) {
        const int y = y0 + r;
        if (y >= H) break;   // uniform
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = 2 * y - 1 + i;
            if (oy < 0 || oy >= OH) continue;
            const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.f);
            const int yy0 = min((int)sy, H - 1), yy1 = min(yy0 + 1, H - 1);
            const float ly = sy - (float)yy0;
            const float wy = (yy0 == y ? 1.f - ly : 0.f) + (yy1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            acc += wy * hs[2 * r + i];
        }
        gin[((size_t)plane * H + y) * W + x] = acc;
    }
}

// ... and for S = 4 (mask_upsample_stride of the shipped video configs: the x4 loss masks of a training stage are 981 MB per 4 frames —
// the generic kernel fetched every element 4 x 4 times with 64 scalar loads per thread: 955 us per stage, 21 % of the x4 training step).
// A thread owns one input column and UB4_R consecutive input rows; per output row it loads the aligned float4 (columns 4x .. 4x + 3) and
// the float2 on either side (4x - 2, 4x - 1 / 4x + 4, 4x + 5: the neighbours' lines), forms the horizontal sum once — adjacent input rows
// share four of their eight output rows — and blends vertically.  Same weights and order of additions as k_upsample_bwd<4>: bit-identical.
#define UB4_R 2
__global__ __launch_bounds__(256) void k_upsample_bwd4(const float* __restrict__ gout, float* __restrict__ gin, int H, int W) {
    const int plane = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * UB4_R;
    if (x >= W) return;   // (whole waves: W % 64 == 0)
    const int OH = 4 * H, OW = 4 * W;
    const float* gp = gout + (size_t)plane * OH * OW;
    float wxs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ox = 4 * x - 2 + j;
        const float sx = fmaxf(((float)ox + 0.5f) * 0.25f - 0.5f, 0.f);
        const int x0 = min((int)sx, W - 1), x1 = min(x0 + 1, W - 1);
        const float lx = sx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        wxs[j] = (ox >= 0 && ox < OW) ? wx : 0.f;
    }
    constexpr int NR = 4 * UB4_R + 4;   // output rows 4 y0 - 2 .. 4 (y0 + R - 1) + 5
    f32x4 vv[NR];
    float2 vm[NR], vp[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {   // ALL loads first, from clamped addresses (see k_upsample_bwd2)
        const int oy = min(max(4 * y0 - 2 + i, 0), OH - 1);
        const float* rp = gp + (size_t)oy * OW + 4 * x;
        vv[i] = *reinterpret_cast<const f32x4*>(rp);
        vm[i] = *reinterpret_cast<const float2*>(rp + (x > 0 ? -2 : 0));
        vp[i] = *reinterpret_cast<const float2*>(rp + (x + 1 < W ? 4 : 2));
    }
    float hs[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int oy = 4 * y0 - 2 + i;
        float h = 0.f;
        if (oy >= 0 && oy < OH) {   // uniform
            if (wxs[0] != 0.f) h += wxs[0] * vm[i].x;
            if (wxs[1] != 0.f) h += wxs[1] * vm[i].y;
            if (wxs[2] != 0.f) h += wxs[2] * vv[i][0];
            if (wxs[3] != 0.f) h += wxs[3] * vv[i][1];
            if (wxs[4] != 0.f) h += wxs[4] * vv[i][2];
            if (wxs[5] != 0.f) h += wxs[5] * vv[i][3];
            if (wxs[6] != 0.f) h += wxs[6] * vp[i].x;
            if (wxs[7] != 0.f) h += wxs[7] * vp[i].y;
        }
        hs[i] = h;
    }
#pragma unroll
    for (int r = 0; r < UB4_R; ++r) {
        const int y = y0 + r;
        if (y >= H) break;   // uniform
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int oy = 4 * y - 2 + i;
            if (oy < 0 || oy >= OH) continue;
            const float sy = fmaxf(((float)oy + 0.5f) * 0.25f - 0.5f, 0.f);
            const int yy0 = min((int)sy, H - 1), yy1 = min(yy0 + 1, H - 1);
            const float ly = sy - (float)yy0;
            const float wy = (yy0 == y ? 1.f - ly : 0.f) + (yy1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            acc += wy * hs[4 * r + i];
        }
        gin[((size_t)plane * H + y) * W + x] = acc;
    }
}


// ---- the training tail (round 5): what `get_targets` + `loss` do per stage as FOUR small kernels around the three mask-loss passes.
// k_stage_targets: one workgroup per image writes that image's rows of labels / label_weights / row_weight / rowk / tgt_row and its
// slice of pos_rows (knet/det/kernel_update_head.py:332-441 — `_get_target_single` per image + the concatenation).
struct TailBatch { VknTailImage img[VKN_TAIL_MAX_IMAGES]; };
__global__ __launch_bounds__(256) void k_stage_targets(const TailBatch batch, int b0, int N, int S, int T, int ncls, int wcols, float pw,
                                                       long long* __restrict__ labels, float* __restrict__ label_weights,
                                                       float* __restrict__ row_weight, int* __restrict__ rowk,
                                                       long long* __restrict__ pos_rows, int* __restrict__ tgt_row,
                                                       int* __restrict__ status) {
    const VknTailImage& im = batch.img[blockIdx.x];
    const int Ns = N + S, tid = threadIdx.x;
    const size_t base = (size_t)(b0 + blockIdx.x) * Ns;
    for (int i = tid; i < Ns; i += 256) {
        labels[base + i] = ncls;     // background
        row_weight[base + i] = 0.f;
        rowk[base + i] = -1;
        tgt_row[base + i] = -1;
    }
    // prediction rows: 1 over the first `wcols` columns (the thing columns when stuff targets exist, :388; else all); stuff rows: the
    // identity over the stuff columns (:391)
    for (int i = tid; i < Ns * ncls; i += 256) {
        const int r = i / ncls, c = i - r * ncls;
        label_weights[base * ncls + i] = r < N ? (c < wcols ? 1.f : 0.f) : ((c >= T && c - T == r - N) ? 1.f : 0.f);
    }
    int bad = 0;
    __syncthreads();
    for (int j = tid; j < im.k; j += 256) {   // matched predictions: ascending rows (vkn_lsap_batch_f32 emits them sorted)
        const int r = im.row_ind[j], g = im.col_ind[j];
        const size_t row = base + r;
        labels[row] = im.gt_labels[g];
        row_weight[row] = 1.f;
        tgt_row[row] = im.gt_row0 + g;
        rowk[row] = im.pos0 + j;
        pos_rows[im.pos0 + j] = (long long)row;
        if (pw != 1.f)
            for (int c = 0; c < wcols; ++c) label_weights[row * ncls + c] = pw;
    }
    for (int j = tid; j < im.n_sem; j += 256) {   // present stuff classes: row N + (class - T), positives in ascending row order
        long long c = im.sem_cls[j];
        int rank = 0;
        for (int i = 0; i < im.n_sem; ++i) {
            const long long ci = im.sem_cls[i];
            rank += (ci < c || (ci == c && i < j)) ? 1 : 0;
            // a class listed twice: the reference keeps the LAST mask and counts the class once (sem_targets[sem_inds] = gt_sem_seg,
            // knet/det/kernel_update_head.py:372-378) — the caller's positive count (k + n_sem, known from shapes) would be one too
            // many and two threads would write one row: reported like an out-of-range class instead of computed wrongly
            if (ci == c && i != j) bad |= 1;
        }
        if (c < T || c >= T + S) { bad |= 1; c = T; }   // out of range: a valid dummy row, reported through `status`
        const size_t row = base + N + (int)(c - T);
        labels[row] = c;
        row_weight[row] = 1.f;
        tgt_row[row] = im.sem_row0 + j;
        rowk[row] = im.pos0 + im.k + rank;
        pos_rows[im.pos0 + im.k + rank] = (long long)row;
    }
    if (bad && status) atomicOr(status, bad);
}

// k_tail_final: ONE workgroup turns the partial sums of k_focal / k_ml_rows / k_ml_rank_fwd into the stage's five outputs
//   out[0] loss_cls = w_cls sum_focal / avg      out[1] pos_acc = 100 #(argmax == label over the K positive rows) / K
//   out[2] loss_mask = w_mask sum_bce / (K P)    out[3] loss_dice = w_dice mean_k(1 - 2 a / ((b + eps) + (c + eps)))
//   out[4] loss_rank = w_rank sum_rank / (B P)   and keeps a, bc per positive row for backward.  Fixed-order sums: deterministic.
__device__ __forceinline__ float tail_block_sum(float v, float* red) {
    v = vkn_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_tail_final(const VknTailCfg cfg, const float* __restrict__ avg_dev,
                                                    const float* __restrict__ focal_partial, int n_focal,
                                                    const float* __restrict__ row_partial, int K, int nchunk,
                                                    const float* __restrict__ rank_partial, int n_rank,
                                                    const float* __restrict__ cls_logits, const long long* __restrict__ labels,
                                                    const long long* __restrict__ pos_rows, int ncls, int B, int P,
                                                    float* __restrict__ out, float* __restrict__ dice_a, float* __restrict__ dice_bc) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float f = 0.f;
    for (int i = tid; i < n_focal; i += 256) f += focal_partial[i];
    f = tail_block_sum(f, red);
    float bce = 0.f, dice = 0.f, hit = 0.f;
    for (int k = tid; k < K; k += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
        for (int c = 0; c < nchunk; ++c) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(row_partial + ((size_t)k * nchunk + c) * 4);
            s0 += q[0]; s1 += q[1]; s2 += q[2]; s3 += q[3];
        }
        const float bc = (s2 + cfg.dice_eps) + (s3 + cfg.dice_eps);
        bce += s0;
        dice += 1.0f - (2.0f * s1) / bc;
        dice_a[k] = s1;
        dice_bc[k] = bc;
        if (cls_logits) {   // top-1 of the row (first maximum) against its label
            const size_t row = (size_t)pos_rows[k];
            const float* z = cls_logits + row * ncls;
            int am = 0;
            float mx = z[0];
            for (int c = 1; c < ncls; ++c)
                if (z[c] > mx) { mx = z[c]; am = c; }
            hit += ((long long)am == labels[row]) ? 1.f : 0.f;
        }
    }
    bce = tail_block_sum(bce, red);
    dice = tail_block_sum(dice, red);
    hit = tail_block_sum(hit, red);
    float rk = 0.f;
    for (int i = tid; i < n_rank; i += 256) rk += rank_partial[i];
    rk = tail_block_sum(rk, red);
    if (tid == 0) {
        const float avg = avg_dev ? avg_dev[0] : cfg.avg_factor;
        out[0] = f * (cfg.w_cls / avg);
        out[1] = hit * (100.0f / (float)K);
        out[2] = cfg.w_mask * (bce / ((float)K * (float)P));
        out[3] = cfg.w_dice * (dice / (float)K);
        out[4] = cfg.with_rank ? cfg.w_rank * (rk / ((float)B * (float)P)) : 0.f;
    }
}

// out = in * (host_scale * g / d), g and d device scalars (NULL = 1): the backward of a loss that is `sum * weight / avg_factor`
__global__ __launch_bounds__(256) void k_scale_by(const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ d,
                                                  float host_scale, float* __restrict__ out, size_t n) {
    const float s = host_scale * (g ? g[0] : 1.f) / (d ? d[0] : 1.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i] * s;
}

// SGD with momentum over one FLAT parameter range (torch.optim.SGD's update, dampening 0, no nesterov):
//   g' = g * grad_scale + weight_decay * p;   m = momentum * m + g';   p -= lr * m          (m starts at zero: the first step gives m = g')
// One pass over the three arrays (16-byte accesses; the tail element-wise).  grad_scale folds the data-parallel mean (1 / world).
__global__ __launch_bounds__(256) void k_sgd_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, size_t n,
                                                  float lr, float momentum, float weight_decay, float grad_scale) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * grad_scale + weight_decay * pv[e];
            mv[e] = momentum * mv[e] + ge;
            pv[e] = pv[e] - lr * mv[e];
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float ge = g[i] * grad_scale + weight_decay * p[i];
        const float mv = momentum * m[i] + ge;
        m[i] = mv;
        p[i] = p[i] - lr * mv;
    }
}

// status |= flag when any of v[0 .. n) lies outside [lo, hi)
__global__ __launch_bounds__(256) void k_check_range(const long long* __restrict__ v, size_t n, long long lo, long long hi, int flag,
                                                     int* __restrict__ status) {
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= (v[i] < lo || v[i] >= hi) ? 1 : 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, flag);
}

// ---- glue of the gather / decode BACKWARD passes (autograd.py), each formerly 2 - 8 element-wise torch launches:
// k_absmax_pow2: s = the power of two that puts max|t| into [2^(tlog2 - 1), 2^tlog2) (what the f16 hi/lo split of the gradient operands
// wants), out[0] = s, out[4] = 1 / s.  One launch: block maxima meet in scratch[0] (atomicMax on the bit pattern of a non-negative
// float), the LAST block (ticket in scratch[1]) finishes and re-zeroes the scratch for the next call on the stream.
__device__ __forceinline__ float tail_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__global__ __launch_bounds__(256) void k_absmax_pow2(const float* __restrict__ t, size_t n, int vec, int tlog2, float* __restrict__ out,
                                                     unsigned* __restrict__ scratch) {
    __shared__ float red[4];
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
        for (size_t i = i0; i < n / 4; i += stride) {
            const f32x4 v = t4[i];
            m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
        for (size_t i = (n / 4) * 4 + i0; i < n; i += stride) m = fmaxf(m, fabsf(t[i]));
    } else {
        for (size_t i = i0; i < n; i += stride) m = fmaxf(m, fabsf(t[i]));
    }
    m = tail_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(&scratch[0], __float_as_uint(m));
        __threadfence();
        if (atomicAdd(&scratch[1], 1u) == gridDim.x - 1) {
            const float mm = __uint_as_float(atomicMax(&scratch[0], 0u));
            int e = 0;
            if (mm > 0.f) (void)frexpf(mm, &e);   // mm = mantissa 2^e, mantissa in [0.5, 1)
            const int sh = min(max(tlog2 - e, -100), 100);
            const float sc = ldexpf(1.0f, sh);
            out[0] = sc;
            out[4] = 1.0f / sc;
            scratch[0] = 0u;
            scratch[1] = 0u;
        }
    }
}

// out [B][Rp][P] = t [B][R][P] * s, rows R .. Rp zero (the decode kernel's 32-row contraction step)
__global__ __launch_bounds__(256) void k_scale_pad_rows(const float* __restrict__ t, const float* __restrict__ sc, int R, int Rp, size_t P,
                                                        int vec, float* __restrict__ out) {
    const int b = blockIdx.y / Rp, r = blockIdx.y - b * Rp;
    const float s = sc ? sc[0] : 1.f;
    float* o = out + ((size_t)b * Rp + r) * P;
    const float* src = t + ((size_t)b * R + r) * P;
    const size_t lo = (size_t)blockIdx.x * 4096, hi = lo + 4096 < P ? lo + 4096 : P;
    if (vec) {
        for (size_t p = lo + 4 * threadIdx.x; p < hi; p += 1024) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < R) { v = *reinterpret_cast<const f32x4*>(src + p); v *= s; }
            *reinterpret_cast<f32x4*>(o + p) = v;
        }
    } else {
        for (size_t p = lo + threadIdx.x; p < hi; p += 256) o[p] = r < R ? src[p] * s : 0.f;
    }
}

// out [B][C][Np] = k [B][N][C]^T * s, columns N .. Np zero
__global__ __launch_bounds__(256) void k_transpose_pad(const float* __restrict__ k, const float* __restrict__ sc, int N, int C, int Np,
                                                       float* __restrict__ out) {
    const int b = blockIdx.y;
    const float s = sc ? sc[0] : 1.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < C * Np; i += gridDim.x * 256) {
        const int c = i / Np, n = i - c * Np;
        out[(size_t)b * C * Np + i] = n < N ? k[((size_t)b * N + n) * C + c] * s : 0.f;
    }
}

// rows [B][Np][P] f16 = (logits [B][N][P] >= thr) ? 1 : 0, rows N .. Np zero: what the gather's backward multiplies with, kept by forward
__global__ __launch_bounds__(256) void k_threshold_rows(const float* __restrict__ z, float thr, int N, int Np, size_t P, int vec,
                                                        _Float16* __restrict__ out) {
    const int b = blockIdx.y / Np, r = blockIdx.y - b * Np;
    _Float16* o = out + ((size_t)b * Np + r) * P;
    const float* src = z + ((size_t)b * N + r) * P;
    const size_t lo = (size_t)blockIdx.x * 4096, hi = lo + 4096 < P ? lo + 4096 : P;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    if (vec) {
        for (size_t p = lo + 4 * threadIdx.x; p < hi; p += 1024) {
            h4 v = {0, 0, 0, 0};
            if (r < N) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(src + p);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = zz[e] >= thr ? (_Float16)1.0f : (_Float16)0.0f;
            }
            *reinterpret_cast<h4*>(o + p) = v;
        }
    } else {
        for (size_t p = lo + threadIdx.x; p < hi; p += 256) o[p] = (r < N && src[p] >= thr) ? (_Float16)1.0f : (_Float16)0.0f;
    }
}

// dk [B][N][C] = dk_p [B][Np][C][:N] * s, dkb [B][N] = dkb_p [B][Np][:N] * s
__global__ __launch_bounds__(256) void k_unscale_rows(const float* __restrict__ dkp, const float* __restrict__ dkbp, const float* __restrict__ sc,
                                                      int N, int Np, int C, float* __restrict__ dk, float* __restrict__ dkb) {
    const int b = blockIdx.y;
    const float s = sc[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * C; i += gridDim.x * 256) dk[(size_t)b * N * C + i] = dkp[(size_t)b * Np * C + i] * s;
    if (blockIdx.x == 0 && dkb)
        for (int n = threadIdx.x; n < N; n += 256) dkb[(size_t)b * N + n] = dkbp[(size_t)b * Np + n] * s;
}

// out = sum of up to VKN_SUM_MAX tensors, one pass (the gradient contributions of the feature map: six per training step, which
// autograd would add pair by pair — five passes of two reads and one write each)
struct SumSrc { const float* p[VKN_SUM_MAX]; };
__global__ __launch_bounds__(256) void k_sum_n(const SumSrc src, int nsrc, size_t n4, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = reinterpret_cast<const f32x4*>(src.p[0])[i];
        for (int k = 1; k < nsrc; ++k) a += reinterpret_cast<const f32x4*>(src.p[k])[i];
        reinterpret_cast<f32x4*>(out)[i] = a;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float a = src.p[0][i];
        for (int k = 1; k < nsrc; ++k) a += src.p[k][i];
        out[i] = a;
    }
}

}  // namespace

extern "C" {

int vkn_upsample_bilinear_bwd_f32(const float* grad_out, float* grad_in, int planes, int H, int W, int S, void* stream) {
    if (!grad_out || !grad_in || planes <= 0 || H <= 0 || W <= 0 || S < 1) return VKN_E_ARG;
    if (S > 4 && S != 8) return VKN_E_SHAPE;   // scale factors of the shipped configs: 2 (training), 4 (inference), 8 (semantic branch)
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (S == 2 && (W % 64) == 0 && ((reinterpret_cast<uintptr_t>(grad_out) & 7) == 0)) {   // the training configs' scale
        for (int done = 0; done < planes; done += 32768) {  // gridDim.z <= 65535
            const int chunk = (planes - done > 32768) ? 32768 : planes - done;
            hipLaunchKernelGGL(k_upsample_bwd2, dim3((W + 255) / 256, (H + UB2_R - 1) / UB2_R, chunk), dim3(256), 0, st,
                               grad_out + (size_t)done * H * S * W * S, grad_in + (size_t)done * H * W, H, W);
            VKN_CHECK_LAUNCH();
        }
        return VKN_OK;
    }
    if (S == 4 && (W % 64) == 0 && ((reinterpret_cast<uintptr_t>(grad_out) & 15) == 0)) {   // the video configs' training scale
        for (int done = 0; done < planes; done += 32768) {
            const int chunk = (planes - done > 32768) ? 32768 : planes - done;
            hipLaunchKernelGGL(k_upsample_bwd4, dim3((W + 255) / 256, (H + UB4_R - 1) / UB4_R, chunk), dim3(256), 0, st,
                               grad_out + (size_t)done * H * S * W * S, grad_in + (size_t)done * H * W, H, W);
            VKN_CHECK_LAUNCH();
        }
        return VKN_OK;
    }
    for (int done = 0; done < planes; done += 32768) {  // gridDim.y <= 65535
        const int chunk = (planes - done > 32768) ? 32768 : planes - done;
        const float* go = grad_out + (size_t)done * H * S * W * S;
        float* gi = grad_in + (size_t)done * H * W;
        const dim3 grid((H * W + 255) / 256, chunk);
        switch (S) {
#define UB_CASE(SV) case SV: hipLaunchKernelGGL(k_upsample_bwd<SV>, grid, dim3(256), 0, st, go, gi, H, W); break;
            UB_CASE(1) UB_CASE(2) UB_CASE(3) UB_CASE(4) UB_CASE(8)
#undef UB_CASE
            default: return VKN_E_SHAPE;
        }
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}


int vkn_focal_loss_blocks(int M, int ncls) {
    if (M <= 0 || ncls <= 0) return 0;
    const size_t nb = ((size_t)M * ncls + 255) / 256;
    return (int)(nb < 512 ? nb : 512);
}

int vkn_focal_loss_f32(const float* logits, const long long* labels, const float* weight, int weight_elementwise, int M, int ncls,
                       float alpha, float gamma, float* partial, float* grad, void* stream) {
    if (!logits || !labels || !partial || !grad || M <= 0 || ncls <= 0) return VKN_E_ARG;
    hipLaunchKernelGGL(k_focal, dim3(vkn_focal_loss_blocks(M, ncls)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels,
                       weight, weight_elementwise, M, ncls, alpha, gamma, partial, grad);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_mask_losses_chunks(int P) { return P > 0 ? (P + ML_CHUNK - 1) / ML_CHUNK : 0; }
int vkn_mask_losses_blocks(int P) { return P > 0 ? (P / ML_PX + 255) / 256 : 0; }   // workgroups per frame of k_ml_rank_fwd

static int ml_fwd(const float* pred, const float* target, const int* tgt_row, const long long* pos_rows, const int* rowk, int K, int B,
                  int Ns, int P, int with_rank, float* row_partial, float* lse, int* top, float* rank_partial, void* stream) {
    if (!pred || !target || B <= 0 || Ns <= 0 || P <= 0 || K < 0) return VKN_E_ARG;
    if ((P & 3) || ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15)) return VKN_E_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (K > 0) {
        if (!pos_rows || !row_partial) return VKN_E_ARG;
        hipLaunchKernelGGL(k_ml_rows, dim3(vkn_mask_losses_chunks(P), K), dim3(256), 0, st, pred, target, tgt_row, pos_rows, P,
                           vkn_mask_losses_chunks(P), row_partial);
        VKN_CHECK_LAUNCH();
    }
    if (with_rank) {
        if (!rowk || !lse || !top || !rank_partial) return VKN_E_ARG;
        hipLaunchKernelGGL(k_ml_rank_fwd, dim3(vkn_mask_losses_blocks(P), B), dim3(256), 0, st, pred, target, tgt_row, rowk, Ns, P, lse,
                           top, rank_partial);
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}

int vkn_mask_losses_fwd_f32(const float* pred, const float* target, const long long* pos_rows, const int* rowk, int K, int B, int Ns,
                            int P, int with_rank, float* row_partial, float* lse, int* top, float* rank_partial, void* stream) {
    return ml_fwd(pred, target, nullptr, pos_rows, rowk, K, B, Ns, P, with_rank, row_partial, lse, top, rank_partial, stream);
}

int vkn_mask_losses_bwd_f32(const float* pred, const float* target, const int* rowk, const float* rowcoef, const float* coef,
                            const float* lse, const int* top, int B, int Ns, int P, int with_rank, float* grad, void* stream) {
    if (!pred || !target || !rowk || !rowcoef || !coef || !grad || B <= 0 || Ns <= 0 || P <= 0) return VKN_E_ARG;
    if (with_rank && (!lse || !top)) return VKN_E_ARG;
    if ((P & 3) || ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad)) & 15))
        return VKN_E_ALIGN;
    MlTail tl = {};
    hipLaunchKernelGGL(k_ml_bwd, dim3((P / 4 + 255) / 256, B, 4), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target, rowk,
                       rowcoef, coef, lse, top, Ns, P, with_rank, grad, tl);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// ---- the training tail (vkn.h)
size_t vkn_sizeof_tail_image(void) { return sizeof(VknTailImage); }
size_t vkn_sizeof_tail_cfg(void) { return sizeof(VknTailCfg); }

int vkn_stage_targets(const VknTailImage* imgs, int B, int N, int S, int T, int ncls, float pos_weight, long long* labels,
                      float* label_weights, float* row_weight, int* rowk, long long* pos_rows, int* tgt_row, int* status, void* stream) {
    if (!imgs || B <= 0 || N <= 0 || S < 0 || T < 0 || ncls <= 0 || !labels || !label_weights || !row_weight || !rowk || !pos_rows || !tgt_row)
        return VKN_E_ARG;
    if (S > 0 && T + S > ncls) return VKN_E_SHAPE;
    const float pw = pos_weight <= 0.f ? 1.f : pos_weight;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int b0 = 0; b0 < B; b0 += VKN_TAIL_MAX_IMAGES) {
        const int nb = (B - b0 < VKN_TAIL_MAX_IMAGES) ? B - b0 : VKN_TAIL_MAX_IMAGES;
        TailBatch batch;
        for (int k = 0; k < nb; ++k) {
            const VknTailImage& im = imgs[b0 + k];
            if (im.k < 0 || im.k > N || im.n_sem < 0 || im.n_sem > (S > 0 ? 4 * S : 0) || (im.k && (!im.row_ind || !im.col_ind || !im.gt_labels)) ||
                (im.n_sem && !im.sem_cls))
                return VKN_E_ARG;
            batch.img[k] = im;
        }
        hipLaunchKernelGGL(k_stage_targets, dim3(nb), dim3(256), 0, st, batch, b0, N, S, T, ncls, S > 0 ? T : ncls, pw, labels, label_weights,
                           row_weight, rowk, pos_rows, tgt_row, status);
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}

int vkn_mask_losses_fwd_bank_f32(const float* pred, const float* bank, const int* tgt_row, const long long* pos_rows, const int* rowk,
                                 int K, int B, int Ns, int P, int with_rank, float* row_partial, float* lse, int* top,
                                 float* rank_partial, void* stream) {
    if (!tgt_row) return VKN_E_ARG;
    return ml_fwd(pred, bank, tgt_row, pos_rows, rowk, K, B, Ns, P, with_rank, row_partial, lse, top, rank_partial, stream);
}

int vkn_mask_losses_lowres_chunks(int h, int w) { return (h > 0 && w > 0) ? ((h + 1) * (w + 1) + 255) / 256 : 0; }

int vkn_mask_losses_fwd_lowres_f32(const float* low, const float* bank, const int* tgt_row, const int* rowk, int K, int B, int Ns, int h,
                                   int w, int S, int with_rank, float* row_partial, float* lse, int* top, float* rank_partial,
                                   void* stream) {
    if (!low || !bank || !tgt_row || !rowk || !row_partial || B <= 0 || Ns <= 0 || h <= 0 || w <= 0 || K <= 0) return VKN_E_ARG;
    if (with_rank && (!lse || !top || !rank_partial)) return VKN_E_ARG;
    if ((S != 2 && S != 4) || Ns > 256 || (size_t)Ns * h * w * sizeof(float) >= (1ull << 31)) return VKN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(bank) | reinterpret_cast<uintptr_t>(lse) | reinterpret_cast<uintptr_t>(top)) & 7) return VKN_E_ALIGN;
    const int nchunk = vkn_mask_losses_lowres_chunks(h, w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (S == 4) hipLaunchKernelGGL(k_ml_fwd_lr<4>, dim3(nchunk, B), dim3(256), 0, st, low, bank, tgt_row, rowk, Ns, h, w, with_rank, row_partial, nchunk, lse, top, rank_partial);
    else hipLaunchKernelGGL(k_ml_fwd_lr<2>, dim3(nchunk, B), dim3(256), 0, st, low, bank, tgt_row, rowk, Ns, h, w, with_rank, row_partial, nchunk, lse, top, rank_partial);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_stage_losses_final_f32(const VknTailCfg* cfg, const float* avg_factor_dev, const float* focal_partial, int n_focal,
                               const float* row_partial, int K, int nchunk, const float* rank_partial, int n_rank,
                               const float* cls_logits, const long long* labels, const long long* pos_rows, int ncls, int B, int P,
                               float* losses, float* dice_a, float* dice_bc, void* stream) {
    if (!cfg || !losses || K <= 0 || !row_partial || !dice_a || !dice_bc || nchunk <= 0 || B <= 0 || P <= 0 || n_focal < 0 || n_rank < 0)
        return VKN_E_ARG;
    if ((n_focal && !focal_partial) || (n_rank && !rank_partial) || (cls_logits && (!labels || !pos_rows || ncls <= 0))) return VKN_E_ARG;
    hipLaunchKernelGGL(k_tail_final, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), *cfg, avg_factor_dev, focal_partial, n_focal,
                       row_partial, K, nchunk, rank_partial, n_rank, cls_logits, labels, pos_rows, ncls, B, P, losses, dice_a, dice_bc);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_mask_losses_bwd_bank_f32(const float* pred, const float* bank, const int* tgt_row, const int* rowk, const float* dice_a,
                                 const float* dice_bc, const float* g_mask, const float* g_dice, const float* g_rank, float w_mask,
                                 float w_dice, float w_rank, int K, const float* lse, const int* top, int B, int Ns, int P,
                                 int with_rank, float* grad, void* stream) {
    if (!pred || !bank || !tgt_row || !rowk || !dice_a || !dice_bc || !grad || B <= 0 || Ns <= 0 || P <= 0 || K <= 0) return VKN_E_ARG;
    if (with_rank && (!lse || !top)) return VKN_E_ARG;
    if ((P & 3) || ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(bank) | reinterpret_cast<uintptr_t>(grad)) & 15))
        return VKN_E_ALIGN;
    MlTail tl = {tgt_row, dice_a, dice_bc, g_mask, g_dice, g_rank, (float)((double)w_mask / ((double)K * (double)P)),
                 (float)((double)w_dice / (double)K), (float)((double)w_rank / ((double)B * (double)P))};
    hipLaunchKernelGGL(k_ml_bwd, dim3((P / 4 + 255) / 256, B, 4), dim3(256), 0, static_cast<hipStream_t>(stream), pred, bank, rowk,
                       (const float*)nullptr, (const float*)nullptr, lse, top, Ns, P, with_rank, grad, tl);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_mask_losses_bwd_lowres_f32(const float* low, const float* bank, const int* tgt_row, const int* rowk, const float* dice_a,
                                   const float* dice_bc, const float* g_mask, const float* g_dice, const float* g_rank, float w_mask,
                                   float w_dice, float w_rank, int K, const float* lse, const int* top, int B, int Ns, int h, int w, int S,
                                   int with_rank, float* grad_low, void* stream) {
    if (!low || !bank || !tgt_row || !rowk || !dice_a || !dice_bc || !grad_low || B <= 0 || Ns <= 0 || h <= 0 || w <= 0 || K <= 0) return VKN_E_ARG;
    if (with_rank && (!lse || !top)) return VKN_E_ARG;
    if (S != 2 && S != 4) return VKN_E_SHAPE;
    const double P = (double)S * h * (double)S * w;
    MlTail tl = {tgt_row, dice_a, dice_bc, g_mask, g_dice, g_rank, (float)((double)w_mask / ((double)K * P)),
                 (float)((double)w_dice / (double)K), (float)((double)w_rank / ((double)B * P))};
    const dim3 grid((w + 62) / 63, (h + LR_NW - 2) / (LR_NW - 1), B * LR_NSPLIT);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (S == 4) hipLaunchKernelGGL(k_ml_bwd_lr<4>, grid, dim3(64 * LR_NW), 0, st, low, bank, rowk, lse, top, Ns, h, w, with_rank, grad_low, tl);
    else hipLaunchKernelGGL(k_ml_bwd_lr<2>, grid, dim3(64 * LR_NW), 0, st, low, bank, rowk, lse, top, Ns, h, w, with_rank, grad_low, tl);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_scale_by_f32(const float* in, const float* g, const float* d, float host_scale, float* out, size_t n, void* stream) {
    if (!in || !out) return VKN_E_ARG;
    if (n == 0) return VKN_OK;
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_scale_by, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), in, g, d,
                       host_scale, out, n);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_sgd_momentum_f32(float* param, const float* grad, float* mom, size_t n, float lr, float momentum, float weight_decay,
                         float grad_scale, void* stream) {
    if (!param || !grad || !mom) return VKN_E_ARG;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(mom)) & 15) return VKN_E_ALIGN;
    if (n == 0) return VKN_OK;
    const size_t nb = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(k_sgd_flat, dim3((unsigned)(nb < 1 ? 1 : (nb < 2048 ? nb : 2048))), dim3(256), 0, static_cast<hipStream_t>(stream), param,
                       grad, mom, n, lr, momentum, weight_decay, grad_scale);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_check_range_i64(const long long* v, size_t n, long long lo, long long hi, int flag, int* status, void* stream) {
    if (!status || (n && !v)) return VKN_E_ARG;
    if (n == 0) return VKN_OK;
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_check_range, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(256), 0, static_cast<hipStream_t>(stream), v, n, lo, hi,
                       flag, status);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}


// ---- glue of the gather / decode backward passes (vkn.h)
int vkn_pow2_scale_f32(const float* t, size_t n, int target_log2, float* scale8, unsigned int* scratch2, void* stream) {
    if (!scale8 || !scratch2 || (n && !t)) return VKN_E_ARG;
    const size_t nb = (n + 4095) / 4096;
    const int vec = (reinterpret_cast<uintptr_t>(t) & 15) == 0;
    hipLaunchKernelGGL(k_absmax_pow2, dim3((unsigned)(nb < 1 ? 1 : (nb < 1024 ? nb : 1024))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       t, n, vec, target_log2, scale8, scratch2);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_scale_pad_rows_f32(const float* t, const float* scale, int B, int R, int Rp, size_t P, float* out, void* stream) {
    if (!t || !out || B <= 0 || R <= 0 || Rp < R || P == 0) return VKN_E_ARG;
    if ((size_t)B * Rp > 65535) return VKN_E_SHAPE;
    const int vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    hipLaunchKernelGGL(k_scale_pad_rows, dim3((unsigned)((P + 4095) / 4096), B * Rp), dim3(256), 0, static_cast<hipStream_t>(stream), t,
                       scale, R, Rp, P, vec, out);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_transpose_pad_f32(const float* k, const float* scale, int B, int N, int C, int Np, float* out, void* stream) {
    if (!k || !out || B <= 0 || N <= 0 || C <= 0 || Np < N) return VKN_E_ARG;
    if (B > 65535) return VKN_E_SHAPE;
    const int nb = (C * Np + 255) / 256;
    hipLaunchKernelGGL(k_transpose_pad, dim3(nb < 64 ? nb : 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), k, scale, N, C, Np, out);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_threshold_rows_f16(const float* logits, float thr_logit, int B, int N, int Np, size_t P, void* rows_f16, void* stream) {
    if (!logits || !rows_f16 || B <= 0 || N <= 0 || Np < N || P == 0) return VKN_E_ARG;
    if ((size_t)B * Np > 65535) return VKN_E_SHAPE;
    const int vec = (P % 4 == 0) && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(rows_f16) & 7) == 0;
    hipLaunchKernelGGL(k_threshold_rows, dim3((unsigned)((P + 4095) / 4096), B * Np), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       thr_logit, N, Np, P, vec, static_cast<_Float16*>(rows_f16));
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_unscale_rows_f32(const float* dk_p, const float* dkb_p, const float* scale, int B, int N, int Np, int C, float* dk, float* dkb,
                         void* stream) {
    if (!dk_p || !scale || !dk || B <= 0 || N <= 0 || Np < N || C <= 0 || (dkb && !dkb_p)) return VKN_E_ARG;
    if (B > 65535) return VKN_E_SHAPE;
    const int nb = (N * C + 255) / 256;
    hipLaunchKernelGGL(k_unscale_rows, dim3(nb < 64 ? nb : 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), dk_p, dkb_p, scale, N, Np,
                       C, dk, dkb);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}


int vkn_sum_n_f32(const float* const* srcs, int nsrc, size_t n, float* out, void* stream) {
    if (!srcs || !out || nsrc <= 0 || nsrc > VKN_SUM_MAX) return VKN_E_ARG;
    if (n == 0) return VKN_OK;
    SumSrc src = {};
    uintptr_t al = reinterpret_cast<uintptr_t>(out);
    for (int k = 0; k < nsrc; ++k) {
        if (!srcs[k]) return VKN_E_ARG;
        src.p[k] = srcs[k];
        al |= reinterpret_cast<uintptr_t>(srcs[k]);
    }
    const size_t n4 = (al & 15) ? 0 : n / 4;
    const size_t nb = ((n4 ? n4 : n) + 255) / 256;
    hipLaunchKernelGGL(k_sum_n, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, static_cast<hipStream_t>(stream), src, nsrc, n4, n, out);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

}  // extern "C"
