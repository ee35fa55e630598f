// vkn_panoptic.hip — post-head mask pipeline, joint panoptic merge.
//
// Reference (per image): KernelIterHead.get_panoptic + merge_stuff_thing_stuff_joint (knet/det/kernel_iter_head.py:332-370,
// 467-524; video: knet/video/kernel_iter_head.py:591-640, 832-905) with KernelUpdateHead.rescale_masks
// (knet/det/kernel_update_head.py:443-458) and the last-stage F.interpolate of _mask_forward (knet/det/kernel_iter_head.py:122-130):
//
//   scaled      = interpolate(mask_logits, scale_factor=up)                      [N, Hm*up, Wm*up]
//   thing rows  = top-`max_per_img` of cls[:Np, :T].flatten()  -> (score, row = idx // T, label = idx % T)
//   stuff rows  = sort(diag(cls[Np:, T:]), descending)         -> (score, row = Np + j, label = T + j)
//   total_masks = interpolate(interpolate(sigmoid(scaled[rows]), size=batch_input_shape)[:, :h, :w], size=ori_shape)
//   cur_mask_ids = argmax_k(score_k * total_masks[k]);  area_k = #(ids == k);  orig_k = #(total_masks[k] >= 0.5)
//   in score order: skip things below instance_score_thr; keep k iff area_k > 0, orig_k > 0, area_k / orig_k >= overlap_thr;
//   panoptic_seg[ids == k] = running segment id.
//
// The reference materialises K x Ho x Wo fp32 (981 MB per 1024x2048 frame) three times.  Here nothing of that size exists:
// one kernel walks 64x16 output tiles, and for a batch of kernels k resamples the tile's footprint level by level through LDS
// (low-res logits -> x`up` + sigmoid -> batch-input size -> ori size), keeping the running arg-max and the two pixel counts; the
// only full-size array is the int32 id map, which the relabel pass turns into panoptic_seg in place.  Integer counts use integer
// atomics (order-independent => deterministic).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

#define PAN_TW 64
#define PAN_TH 16
#define PAN_PPT (PAN_TH / 4)  // output pixels per thread: rows fy * PAN_PPT + j of one column (adjacent rows share their input rows)
#define PAN_THREADS 256
#define PAN_KB 4

// ATen's linear-interpolation coefficients, align_corners=False (aten/src/ATen/native/UpSample.h:
// area_pixel_compute_source_index + guard_index_and_lambda), fp32 opmath.
__device__ __forceinline__ void pan_coef(float scale, int dst, int in_size, int& i0, int& i1, float& lam) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = min((int)floorf(src), in_size - 1);
    lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
}

// value of one output pixel of a level from its input region in LDS (row pitch `pitch`), ATen's evaluation order:
// x first, then y; each as v0*w0 + v1*w1.  Region buffers carry one replicated column and row past their extent, so the
// second tap is always at +1 (ATen clamps it onto the first at the image border — same value, same arithmetic) and the two
// taps of a row are one ds_read2.
__device__ __forceinline__ float pan_lerp(const float* __restrict__ ldsf, int base, int pitch, int y0, float ly, int x0,
                                          float lx) {
    // `ldsf` is the dynamic-LDS base itself and every buffer is addressed by a word offset: a pointer selected at run time
    // (level 1 or 2 buffer) loses its address space and turns these reads into flat loads (measured: 2.5x slower)
    const int o = base + y0 * pitch + x0;
    const float wx0 = 1.f - lx, wy0 = 1.f - ly;
    const float r0 = ldsf[o] * wx0 + ldsf[o + 1] * lx;
    const float r1 = ldsf[o + pitch] * wx0 + ldsf[o + pitch + 1] * lx;
    return r0 * wy0 + r1 * ly;
}

// ------------------------------------------------------------------------------------------------ selection
// Ranks by (score descending, index ascending) — torch.topk / sort / argsort leave the order of exact
// ties unspecified; this is the stable choice.  sel_* [B][K], K = Kt + nstuff (the merge order is ranked in k_pan_merge).
// grid (chunks, B): every workgroup stages all candidates of its frame in LDS and ranks its own slice of them (rank = number of
// candidates that sort before it) — O(n^2 / workgroups) per workgroup, n = Np * T up to 8000 (COCO: 100 proposals x 80 classes).
__global__ __launch_bounds__(256) void k_pan_select(const float* __restrict__ cls, int N, int ncls, int Np, int T, int Kt,
                                                    int nstuff, int* __restrict__ sel_row, int* __restrict__ sel_label,
                                                    float* __restrict__ sel_score) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, K = Kt + nstuff;
    const float* c = cls + (size_t)b * N * ncls;
    int* row = sel_row + (size_t)b * K;
    int* lab = sel_label + (size_t)b * K;
    float* sc = sel_score + (size_t)b * K;
    const int nth = Np * T;
    float* cand = reinterpret_cast<float*>(smem);  // [nth] thing candidates, then [nstuff] stuff candidates
    for (int i = threadIdx.x; i < nth; i += 256) cand[i] = c[(size_t)(i / T) * ncls + (i % T)];
    for (int i = threadIdx.x; i < nstuff; i += 256) cand[nth + i] = c[(size_t)(Np + i) * ncls + (T + i)];
    __syncthreads();
    // things: candidate i = (proposal i / T, class i % T)                                knet/det/kernel_iter_head.py:334-340
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nth; i += gridDim.x * 256) {
        const float s = cand[i];
        int rank = 0;
        for (int j = 0; j < nth; ++j) {
            const float t = cand[j];
            rank += (t > s) || (t == s && j < i);
        }
        if (rank < Kt) {
            row[rank] = i / T;
            lab[rank] = i % T;
            sc[rank] = s;
        }
    }
    // stuff: score j = cls[Np + j][T + j], sorted descending; joint labels = T + j         :349-352, :359
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nstuff; i += gridDim.x * 256) {
        const float s = cand[nth + i];
        int rank = 0;
        for (int j = 0; j < nstuff; ++j) {
            const float t = cand[nth + j];
            rank += (t > s) || (t == s && j < i);
        }
        row[Kt + rank] = Np + i;
        lab[Kt + rank] = T + i;
        sc[Kt + rank] = s;
    }
}

// ------------------------------------------------------------------------------------------------ fused resample + arg-max
struct PanLevel {
    int in_h, in_w;  // size of the level's input image
    float sy, sx;    // output -> input coordinate scale
};
struct PanGeom {
    int Hm, Wm;       // low-res logits
    PanLevel lv[3];   // [0] x`up` of the logits (sigmoid applied to its output), [1] -> batch_input_shape, [2] crop -> ori_shape
    int nlev;         // 3, or 2 when the last resize is the identity (ori_shape == img_shape)
    int Ho, Wo;       // output size
    int cap_w[3], cap_h[3];  // LDS capacity of the INPUT region of level i
};

struct PanTab {  // per-level coefficient tables in LDS (local indices into the level's input region)
    int *x0, *x1, *y0, *y1;
    float *lx, *ly;
};

// Input region of every level for one output tile, from the output back to the logits: a level's input region is spanned by the first
// tap of the first output pixel and the second tap of the last one (the coordinate maps are monotone).  The SAME function gives
// k_pan_bounds its footprints and k_pan_argmax its staging regions — the bounds cover exactly the logits the tile reads.
struct PanRegion { int rx0[3], rw[3], ry0[3], rh[3]; };
__device__ __forceinline__ void pan_region(const PanGeom& g, int X0, int tw, int Y0, int th, PanRegion& R) {
    int ox0 = X0, ow = tw, oy0 = Y0, oh = th;
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        if (l >= g.nlev) continue;
        int a0, a1, b0, b1;
        float lam;
        pan_coef(g.lv[l].sx, ox0, g.lv[l].in_w, a0, a1, lam);
        pan_coef(g.lv[l].sx, ox0 + ow - 1, g.lv[l].in_w, b0, b1, lam);
        R.rx0[l] = a0; R.rw[l] = b1 - a0 + 1;
        pan_coef(g.lv[l].sy, oy0, g.lv[l].in_h, a0, a1, lam);
        pan_coef(g.lv[l].sy, oy0 + oh - 1, g.lv[l].in_h, b0, b1, lam);
        R.ry0[l] = a0; R.rh[l] = b1 - a0 + 1;
        ox0 = R.rx0[l]; ow = R.rw[l]; oy0 = R.ry0[l]; oh = R.rh[l];
    }
}

// ------------------------------------------------------------------------------------------------ footprint bounds (round 6)
// bounds[b][ty][tx][k] = (min, max) of the logits of selected kernel k over the footprint of output tile (tx, ty): ONE coalesced pass
// over the K selected planes (each row is read by the two or three tile rows whose footprints hold it: cache hits) instead of every
// tile workgroup fetching the footprint of all K kernels (4.5x read over-fetch, eight dependent global round trips and a 55-element
// serial min / max walk per kernel and tile: what made k_pan_argmax a 10 us-per-tile latency chain in round 5).
// Workgroup = (PANB_KCH kernels) x (a strip of PANB_TR tile rows) x frame; thread = logit column: column-wise min / max over the tile
// row's footprint rows -> LDS, then one (tile, kernel) pair per thread takes the min / max over the tile's footprint columns.
#define PANB_KCH 2
#define PANB_TR 8
typedef float pan_f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pan_bounds(PanGeom g, const float* __restrict__ masks, const int* __restrict__ sel_row, int K,
                                                    int N, int ntx, int nty, pan_f2* __restrict__ bounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pan_f2* cmm = reinterpret_cast<pan_f2*>(smem);      // [PANB_TR tile rows][PANB_KCH][Wm] column (min, max) over the tile row's footprint rows
    const int tid = threadIdx.x, b = blockIdx.z, k0 = blockIdx.x * PANB_KCH, strip = blockIdx.y;
    const size_t plane = (size_t)g.Hm * g.Wm;
    const float* pl[PANB_KCH];
#pragma unroll
    for (int kk = 0; kk < PANB_KCH; ++kk)
        pl[kk] = masks + ((size_t)b * N + sel_row[(size_t)b * K + min(k0 + kk, K - 1)]) * plane;
    const int ty0 = strip * PANB_TR, ntr = min(nty - ty0, PANB_TR);
    // phase A: every tile row of the strip, no barrier in between; FOUR tile rows' loads (4 x 6 rows x PANB_KCH planes) are requested
    // before the first use — a wave that waits on 12 loads at a time was the whole cost of this kernel (5120 resident waves x 12 loads
    // per ~2 us round trip = 185 us for the 5.7 M wave-loads)
    for (int t0 = 0; t0 < ntr; t0 += 4) {
        int ry0[4], lh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int Y0 = (ty0 + min(t0 + u, ntr - 1)) * PAN_TH, th = min(PAN_TH, g.Ho - Y0);
            PanRegion R;
            pan_region(g, 0, 1, Y0, th, R);   // (the row footprint does not depend on the tile column)
            ry0[u] = R.ry0[0]; lh[u] = R.rh[0];
        }
        const int lhm = max(max(lh[0], lh[1]), max(lh[2], lh[3]));
        for (int x = tid; x < g.Wm; x += 256) {
            float mn[4][PANB_KCH], mx[4][PANB_KCH];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int kk = 0; kk < PANB_KCH; ++kk) { mn[u][kk] = INFINITY; mx[u][kk] = -INFINITY; }
            for (int y6 = 0; y6 < lhm; y6 += 6) {   // (rows past a footprint: its last row again)
                float v[4][6][PANB_KCH];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int yy = 0; yy < 6; ++yy) {
                        const size_t o = (size_t)(ry0[u] + min(y6 + yy, lh[u] - 1)) * g.Wm + x;
#pragma unroll
                        for (int kk = 0; kk < PANB_KCH; ++kk) v[u][yy][kk] = pl[kk][o];
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int yy = 0; yy < 6; ++yy)
#pragma unroll
                        for (int kk = 0; kk < PANB_KCH; ++kk) { mn[u][kk] = fminf(mn[u][kk], v[u][yy][kk]); mx[u][kk] = fmaxf(mx[u][kk], v[u][yy][kk]); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < ntr)
#pragma unroll
                    for (int kk = 0; kk < PANB_KCH; ++kk) cmm[((size_t)(t0 + u) * PANB_KCH + kk) * g.Wm + x] = pan_f2{mn[u][kk], mx[u][kk]};
        }
    }
    __syncthreads();
    // phase B: one (tile row, tile column, kernel) triple per thread
    for (int i = tid; i < ntr * ntx * PANB_KCH; i += 256) {
        const int kk = i % PANB_KCH, tx = (i / PANB_KCH) % ntx, t = i / (PANB_KCH * ntx);
        const int Y0 = (ty0 + t) * PAN_TH, th = min(PAN_TH, g.Ho - Y0);
        const int X0 = tx * PAN_TW, tw = min(PAN_TW, g.Wo - X0);
        PanRegion R;
        pan_region(g, X0, tw, Y0, th, R);
        float mn = INFINITY, mx = -INFINITY;
        const pan_f2* c = cmm + ((size_t)t * PANB_KCH + kk) * g.Wm + R.rx0[0];
        for (int xx = 0; xx < R.rw[0]; ++xx) {
            mn = fminf(mn, c[xx][0]);
            mx = fmaxf(mx, c[xx][1]);
        }
        if (k0 + kk < K) bounds[(((size_t)b * nty + ty0 + t) * ntx + tx) * K + k0 + kk] = pan_f2{mn, mx};
    }
}

// Workgroup = one PAN_TW x PAN_TH output tile of one frame (thread = PAN_PPT output pixels of one column).
//  1. every thread derives the tile's level regions itself (pan_region: no serial section), the coefficient tables of every level are
//     filled in one go — one barrier;
//  2. thread k reads kernel k's footprint bounds (one coalesced load of the tile's K (min, max) pairs, k_pan_bounds).  Bilinear
//     resampling and the sigmoid are monotone convex combinations, so every output value of kernel k in this tile lies in
//     [sigmoid(min_k), sigmoid(max_k)]:  k can neither win a pixel nor reach prob 0.5 here if
//         score_k * sigmoid(max_k) < LB := max_j score_j * sigmoid(min_j)   and   sigmoid(max_k) < 0.5
//     (bounds padded by 1e-5 relative: fp32 rounding of the logits moves a probability by <= 2e-6 relative).  Only the surviving
//     kernels — typically a handful per tile for real segmentation masks — are staged and resampled; the result is identical to
//     visiting all K;
//  3. the survivors' logits footprints -> LDS, every load of a chunk of KC survivors in flight at once (one global round trip);
//  4. per batch of PAN_KB survivors: wave kk resamples kernel kk's footprint level by level through LDS; then every thread
//     evaluates the last level at its output pixels, keeps the running arg-max (strict >, ascending k: first maximum wins,
//     as torch.argmax) and counts prob >= 0.5.
__global__ __launch_bounds__(PAN_THREADS) void k_pan_argmax(PanGeom g, const float* __restrict__ masks,
                                                            const int* __restrict__ sel_row,
                                                            const float* __restrict__ sel_score, int K, int N,
                                                            int* __restrict__ ids, int* __restrict__ area,
                                                            int* __restrict__ orig, int* __restrict__ err, int prune, int KC,
                                                            const pan_f2* __restrict__ bounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int nact_s;
    __shared__ int lb_bits;
    const int tid = threadIdx.x, b = blockIdx.z, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int X0 = blockIdx.x * PAN_TW, Y0 = blockIdx.y * PAN_TH;
    const int nl = g.nlev;
    const int tw = min(PAN_TW, g.Wo - X0), th = min(PAN_TH, g.Ho - Y0);

    // ---- carve LDS: tables, per-kernel scalars, region buffers
    float* const ldsf = reinterpret_cast<float*>(smem);
    int* const ldsi = reinterpret_cast<int*>(smem);
    int woff = 0;  // running word offset of the carve
    auto take_i = [&](int n) { int* r = ldsi + woff; woff += n; return r; };
    auto take_f = [&](int n) { float* r = ldsf + woff; woff += n; return r; };
    auto take_o = [&](int n) { const int r = woff; woff += n; return r; };
    PanTab tab[3] = {};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l >= nl) break;
        const int nw = (l == nl - 1) ? PAN_TW : g.cap_w[l + 1], nh = (l == nl - 1) ? PAN_TH : g.cap_h[l + 1];
        tab[l].x0 = take_i(nw); tab[l].x1 = take_i(nw); tab[l].lx = take_f(nw);
        tab[l].y0 = take_i(nh); tab[l].y1 = take_i(nh); tab[l].ly = take_f(nh);
    }
    int* area_s = take_i(K);
    int* orig_s = take_i(K);
    int* list = take_i(K);     // surviving kernels, ascending
    float* khi = take_f(K);    // score * sigmoid(max logit)  (padded up)
    float* kpm = take_f(K);    // sigmoid(max logit)          (padded up)
    int* srow = take_i(K);     // mask row of kernel k   } read once per tile: the staging and the arg-max loops index them by survivor,
    float* ssc = take_f(K);    // score of kernel k      } which from global memory were two more dependent round trips per tile
    const int ln = g.cap_w[0] * g.cap_h[0], lp = g.cap_w[0];
    const int Lo = take_o(KC * ln);  // word offset of the logits footprints of one chunk of KC survivors
    float* const Ls = ldsf + Lo;
    const int b1o = take_o(PAN_KB * g.cap_w[1] * g.cap_h[1]);
    const int b2o = (nl == 3) ? take_o(PAN_KB * g.cap_w[2] * g.cap_h[2]) : 0;

    for (int i = tid; i < K; i += PAN_THREADS) { area_s[i] = 0; orig_s[i] = 0; }
    if (tid == 0) lb_bits = 0;
    // the tile's only independent global reads — kernel tid's footprint bounds, score and mask row — are requested first: their
    // latency runs under the region / table arithmetic below
    const int* rowp = sel_row + (size_t)b * K;
    const float* scp = sel_score + (size_t)b * K;
    const pan_f2* bt = bounds + (((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * K;
    const int kpre = min(tid, K - 1);
    const pan_f2 bd_pre = bt[kpre];
    const float s_pre = scp[kpre];
    const int row_pre = rowp[kpre];

    // ---- 1. regions (registers, every thread) and coefficient tables (LDS, local indices) of every level
    PanRegion R;
    pan_region(g, X0, tw, Y0, th, R);
    {
        bool bad = false;
#pragma unroll
        for (int l = 0; l < 3; ++l)
            if (l < nl) bad |= (R.rw[l] + 1 > g.cap_w[l]) || (R.rh[l] + 1 > g.cap_h[l]);
        if (bad) {   // capacity bug: reported through `err`, never silent (uniform over the workgroup)
            if (tid == 0) atomicExch(err, 1);
            return;
        }
    }
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        if (l >= nl) continue;
        const int ox0 = (l == nl - 1) ? X0 : R.rx0[l + 1], oy0 = (l == nl - 1) ? Y0 : R.ry0[l + 1];
        const int ow = (l == nl - 1) ? tw : R.rw[l + 1], oh = (l == nl - 1) ? th : R.rh[l + 1];
        for (int i = tid; i < ow; i += PAN_THREADS) {
            int i0, i1;
            float lam;
            pan_coef(g.lv[l].sx, ox0 + i, g.lv[l].in_w, i0, i1, lam);
            tab[l].x0[i] = i0 - R.rx0[l]; tab[l].x1[i] = i1 - R.rx0[l]; tab[l].lx[i] = lam;
        }
        for (int i = tid; i < oh; i += PAN_THREADS) {
            int i0, i1;
            float lam;
            pan_coef(g.lv[l].sy, oy0 + i, g.lv[l].in_h, i0, i1, lam);
            tab[l].y0[i] = i0 - R.ry0[l]; tab[l].y1[i] = i1 - R.ry0[l]; tab[l].ly[i] = lam;
        }
    }

    // ---- 2. bounds of all K kernels over this tile's footprint, survivor list
    {
        float lo_w = 0.f;
        for (int k = tid; k < K; k += PAN_THREADS) {
            const bool pre = k == tid;
            const pan_f2 bd = pre ? bd_pre : bt[k];
            const float s = pre ? s_pre : scp[k];
            srow[k] = pre ? row_pre : rowp[k];
            ssc[k] = s;
            const float pm = 1.0f / (1.0f + expf(-bd[1])), pn = 1.0f / (1.0f + expf(-bd[0]));
            khi[k] = s * pm * (1.0f + 1e-5f) + 1e-30f;
            kpm[k] = pm * (1.0f + 1e-5f);
            lo_w = fmaxf(lo_w, fmaxf(s * pn * (1.0f - 1e-5f), 0.f));
        }
        lo_w = vkn_wave_max(lo_w);
        __syncthreads();   // (lb_bits = 0 and the tables are visible)
        if (lane == 0 && lo_w > 0.f) atomicMax(&lb_bits, __float_as_int(lo_w));  // lo >= 0: the int order of the bits is the float order
        __syncthreads();
    }
    if (wave == 0) {
        const float LB = __int_as_float(lb_bits);
        int cnt = 0;
        for (int base = 0; base < K; base += 64) {
            const int k = base + lane;
            const bool act = k < K && (!prune || khi[k] >= LB || kpm[k] >= 0.5f);
            const unsigned long long m = __ballot(act);
            if (act) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = k;
            cnt += __popcll(m);
        }
        if (lane == 0) nact_s = cnt;
    }
    __syncthreads();
    const int nact = nact_s;

    const int lw = R.rw[0], lh = R.rh[0];
    const int lw1 = lw + 1, lwh = lw1 * (lh + 1);  // staged with the replicated column / row
    const float inv_lw = 1.0f / (float)lw1, inv_lwh = 1.0f / (float)lwh;
    const float* lbase = masks + (size_t)b * N * g.Hm * g.Wm + (size_t)R.ry0[0] * g.Wm + R.rx0[0];
    const size_t plane = (size_t)g.Hm * g.Wm;

    // this thread's output pixels: (fx, fy * PAN_PPT + j), j < PAN_PPT — vertically adjacent, so that consecutive pixels share input rows
    // of the last level (x2 up-scaling: four output rows read three input rows); fy is the wave index: the row pattern is wave-uniform
    const int fx = tid & 63, fy = wave;
    const bool okx = fx < tw;
    const PanTab tf = (nl == 3) ? tab[2] : tab[1];
    const int fip = (nl == 3) ? g.cap_w[2] : g.cap_w[1];
    const float flx = tf.lx[okx ? fx : 0];
    const int fxo = tf.x0[okx ? fx : 0];
    bool okp[PAN_PPT];
    int fy0[PAN_PPT];    // first input row of the pixel (wave-uniform: SGPR)
    float fly[PAN_PPT];
    float best[PAN_PPT];
    int bid[PAN_PPT];
#pragma unroll
    for (int j = 0; j < PAN_PPT; ++j) {
        okp[j] = okx && (fy * PAN_PPT + j) < th;
        const int yy = min(fy * PAN_PPT + j, th - 1);
        fy0[j] = __builtin_amdgcn_readfirstlane(tf.y0[yy]);
        fly[j] = tf.ly[yy];
        best[j] = -INFINITY;
        bid[j] = 0;
    }

    for (int c0 = 0; c0 < nact; c0 += KC) {
        const int nc = min(KC, nact - c0);
        // ---- 3. footprints of this chunk's survivors -> LDS: item i = (survivor i / lwh, element i % lwh), four items per thread in
        //         flight (all of a typical chunk: a handful of survivors x ~70 elements)
        if (c0) __syncthreads();   // the previous chunk's last batch is done with Ls
        const int items = nc * lwh;
        for (int i0 = tid; i0 < items; i0 += 4 * PAN_THREADS) {
            float v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * PAN_THREADS, items - 1);
                const int a = (int)(((float)i + 0.5f) * inv_lwh), r = i - a * lwh;
                const int yy = (int)(((float)r + 0.5f) * inv_lw), xx = r - yy * lw1;
                v[u] = lbase[(size_t)srow[list[c0 + a]] * plane + (size_t)min(yy, lh - 1) * g.Wm + min(xx, lw - 1)];
                dst[u] = a * ln + yy * lp + xx;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * PAN_THREADS < items) Ls[dst[u]] = v[u];
        }
        __syncthreads();

        // ---- 4. survivors of the chunk, PAN_KB at a time: wave kk resamples kernel list[c0 + a0 + kk] through the intermediate levels
        for (int a0 = 0; a0 < nc; a0 += PAN_KB) {
            const int nk = min(PAN_KB, nc - a0);
            for (int kq = wave; kq < nk; kq += PAN_THREADS / 64) {
                const int slot = a0 + kq;
                {   // level 0: logits footprint -> x up -> sigmoid                                     (rescale_masks :446)
                    const int ow = R.rw[1], oh = R.rh[1], op = g.cap_w[1], ow1 = ow + 1;
                    const float inv_ow = 1.0f / (float)ow1;
                    const int so = Lo + slot * ln, dof = b1o + kq * (g.cap_w[1] * g.cap_h[1]);
                    const PanTab& t = tab[0];
                    for (int r = lane; r < ow1 * (oh + 1); r += 64) {
                        const int yy = (int)(((float)r + 0.5f) * inv_ow), xx = r - yy * ow1;
                        const int ys = min(yy, oh - 1), xs = min(xx, ow - 1);
                        const float v = pan_lerp(ldsf, so, lp, t.y0[ys], t.ly[ys], t.x0[xs], t.lx[xs]);
                        ldsf[dof + yy * op + xx] = 1.0f / (1.0f + expf(-v));
                    }
                }
                if (nl == 3) {  // level 1: -> batch_input_shape (the crop is the domain of level 2)
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its own reads
                    __builtin_amdgcn_wave_barrier();
                    const int ow = R.rw[2], oh = R.rh[2], op = g.cap_w[2], ip = g.cap_w[1], ow1 = ow + 1;
                    const float inv_ow = 1.0f / (float)ow1;
                    const int so = b1o + kq * (g.cap_w[1] * g.cap_h[1]), dof = b2o + kq * (g.cap_w[2] * g.cap_h[2]);
                    const PanTab& t = tab[1];
                    for (int r = lane; r < ow1 * (oh + 1); r += 64) {
                        const int yy = (int)(((float)r + 0.5f) * inv_ow), xx = r - yy * ow1;
                        const int ys = min(yy, oh - 1), xs = min(xx, ow - 1);
                        ldsf[dof + yy * op + xx] = pan_lerp(ldsf, so, ip, t.y0[ys], t.ly[ys], t.x0[xs], t.lx[xs]);
                    }
                }
            }
            __syncthreads();
            // ---- last level per output pixel + score-weighted arg-max + ">= 0.5" count                      :484-486, :499
            {
                const int in = fip * ((nl == 3) ? g.cap_h[2] : g.cap_h[1]);
                const int lasto = (nl == 3) ? b2o : b1o;
                for (int kk = 0; kk < nk; ++kk) {
                    const int k = list[c0 + a0 + kk];
                    const int so = lasto + kk * in + fxo;
                    const float s = ssc[k];
                    int c = 0;
                    // ATen's order — x first (one horizontal lerp per input row: v0 * (1 - lx) + v1 * lx), then y — with every input row's
                    // horizontal lerp computed ONCE per thread: the two most recent rows stay in registers (scalar row compares)
                    const float wx0 = 1.f - flx;
                    int ya = -2, yb = -2;
                    float ra = 0.f, rb = 0.f;
                    auto hl = [&](int y) { const int o = so + y * fip; return ldsf[o] * wx0 + ldsf[o + 1] * flx; };
#pragma unroll
                    for (int j = 0; j < PAN_PPT; ++j) {
                        const int y0 = fy0[j];
                        float r0, r1;
                        if (y0 == ya) r0 = ra; else if (y0 == yb) r0 = rb; else r0 = hl(y0);
                        if (y0 + 1 == yb) r1 = rb; else if (y0 + 1 == ya) r1 = ra; else r1 = hl(y0 + 1);
                        ya = y0; ra = r0; yb = y0 + 1; rb = r1;
                        const float v = r0 * (1.f - fly[j]) + r1 * fly[j];
                        const float pj = s * v;
                        if (pj > best[j]) { best[j] = pj; bid[j] = k; }
                        c += __popcll(__ballot(okp[j] && v >= 0.5f));
                    }
                    if (lane == 0 && c) atomicAdd(&orig_s[k], c);
                }
            }
            __syncthreads();
        }
    }

    int* idp = ids + (size_t)b * g.Ho * g.Wo;
#pragma unroll
    for (int j = 0; j < PAN_PPT; ++j)
    {
        if (okp[j]) idp[(size_t)(Y0 + fy * PAN_PPT + j) * g.Wo + X0 + fx] = bid[j];
        // pixel counts: a row of 64 pixels usually has ONE winner — one LDS atomic per wave instead of 64 on the same address
        const int first = __builtin_amdgcn_readfirstlane(bid[j]);
        const unsigned long long act = __ballot(okp[j]), same = __ballot(okp[j] && bid[j] == first);
        if (same == act) {
            if (lane == 0 && act) atomicAdd(&area_s[first], (int)__popcll(act));
        } else if (okp[j]) {
            atomicAdd(&area_s[bid[j]], 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < K; i += PAN_THREADS) {
        if (area_s[i]) atomicAdd(&area[(size_t)b * K + i], area_s[i]);
        if (orig_s[i]) atomicAdd(&orig[(size_t)b * K + i], orig_s[i]);
    }
}

// ------------------------------------------------------------------------------------------------ sequential merge
// One thread per frame: the score-ordered accept / reject loop                       knet/det/kernel_iter_head.py:492-522
// info[B][K][6] = {mask row, joint label, segment id (0 = rejected), area, original area, score bits}
__global__ __launch_bounds__(64) void k_pan_merge(const int* __restrict__ sel_row, const int* __restrict__ sel_label,
                                                  const float* __restrict__ sel_score, int* __restrict__ order,
                                                  const int* __restrict__ area, const int* __restrict__ orig, int K, int T,
                                                  float inst_thr, double overlap_thr, int* __restrict__ seg_of,
                                                  int* __restrict__ info, int* __restrict__ nseg, const int* __restrict__ err,
                                                  int* __restrict__ bbox) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;  // one wave per frame: stage the K-entry tables in LDS, then one lane walks them in score order
    int* ord = reinterpret_cast<int*>(smem);
    int* lab = ord + K;
    int* ar = lab + K;
    int* og = ar + K;
    int* sid_s = og + K;
    float* scs = reinterpret_cast<float*>(sid_s + K);
    for (int i = threadIdx.x; i < K; i += 64) {
        const size_t kk = (size_t)b * K + i;
        lab[i] = sel_label[kk]; ar[i] = area[kk]; og[i] = orig[kk]; scs[i] = sel_score[kk];
    }
    __syncthreads();
    // merge order: argsort(-total_scores), ties by index                                                    :489
    for (int i = threadIdx.x; i < K; i += 64) {
        const float s = scs[i];
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const float t = scs[j];
            rank += (t > s) || (t == s && j < i);
        }
        ord[rank] = i;
        order[(size_t)b * K + rank] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cur = 0;
        for (int r = 0; r < K; ++r) {
            const int k = ord[r];
            int sid = 0;
            if (!(lab[k] < T && scs[k] < inst_thr)) {
                const int a = ar[k], o = og[k];
                if (a > 0 && o > 0 && !((double)a / (double)o < overlap_thr)) sid = ++cur;
            }
            sid_s[k] = sid;
        }
        nseg[b] = *err ? -1 : cur;  // -1: the arg-max kernel hit an LDS capacity bug (never silent)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 64) {
        const size_t kk = (size_t)b * K + i;
        seg_of[kk] = sid_s[i];
        int* e = info + kk * 6;
        e[0] = sel_row[kk]; e[1] = lab[i]; e[2] = sid_s[i]; e[3] = ar[i]; e[4] = og[i]; e[5] = __float_as_int(scs[i]);
        if (bbox) {  // accepted: identity of min / max, filled by the relabel pass; rejected: unitrack's empty box (-1, -1, 10, 10)
            int* bb = bbox + kk * 4;
            const bool acc = sid_s[i] > 0;
            bb[0] = acc ? 0x7fffffff : -1; bb[1] = acc ? 0x7fffffff : -1; bb[2] = acc ? -1 : 10; bb[3] = acc ? -1 : 10;
        }
    }
}

// panoptic_seg[p] = segment id of the kernel that won pixel p (in place over the id map)                :503
// bbox != NULL: also the bounding box (xmin, ymin, xmax, ymax) of every accepted segment = `tensor_mask2box(panoptic_seg == id)`
// (unitrack/utils/mask.py:41-46, 80-90; what the video detector feeds its tracker,
// knet/video/knet_quansi_dense_embed_fc_joint_train.py:541-584), by integer min / max atomics (order-independent).
__global__ __launch_bounds__(256) void k_pan_relabel(int* __restrict__ seg, const int* __restrict__ seg_of, int K, int Wo, size_t npx,
                                                     int* __restrict__ bbox) {
    extern __shared__ int bbs[];  // [K][4]
    const int b = blockIdx.y;
    const int* tbl = seg_of + (size_t)b * K;
    int* s = seg + (size_t)b * npx;
    if (bbox)
        for (int i = threadIdx.x; i < K; i += 256) { bbs[4 * i] = 0x7fffffff; bbs[4 * i + 1] = 0x7fffffff; bbs[4 * i + 2] = -1; bbs[4 * i + 3] = -1; }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const int k = s[i];
        const int sid = tbl[k];
        s[i] = sid;
        if (bbox && sid > 0) {
            const int y = (int)(i / (size_t)Wo), x = (int)(i - (size_t)y * Wo);
            atomicMin(&bbs[4 * k], x); atomicMin(&bbs[4 * k + 1], y);
            atomicMax(&bbs[4 * k + 2], x); atomicMax(&bbs[4 * k + 3], y);
        }
    }
    if (!bbox) return;
    __syncthreads();
    int* gb = bbox + (size_t)b * K * 4;
    for (int i = threadIdx.x; i < K; i += 256)
        if (bbs[4 * i + 2] >= 0) {
            atomicMin(&gb[4 * i], bbs[4 * i]); atomicMin(&gb[4 * i + 1], bbs[4 * i + 1]);
            atomicMax(&gb[4 * i + 2], bbs[4 * i + 2]); atomicMax(&gb[4 * i + 3], bbs[4 * i + 3]);
        }
}

// ------------------------------------------------------------------------------------------------ host side
static int cap_of(int out_extent, float scale, int in_size) {
    // n consecutive outputs span (n-1)*scale input coordinates -> at most that + 3 input indices
    long long c = (long long)ceil((double)(out_extent > 0 ? out_extent - 1 : 0) * (double)scale) + 3;
    if (c > in_size) c = in_size;
    return (int)(c < 1 ? 1 : c) + 1;  // + the replicated border column / row
}

static size_t pan_tables_bytes(int B, int K) {
    // sel_row, sel_label, order, area, orig, seg_of (int) + sel_score (float) + err
    return ((size_t)B * K * 7 * 4 + 256 + 255) & ~(size_t)255;
}
size_t vkn_panoptic_ws_bytes(int B, int K, int Ho, int Wo) {
    // ... + the footprint bounds [B][tiles][K] (min, max)
    const size_t tiles = (size_t)((Wo + PAN_TW - 1) / PAN_TW) * (size_t)((Ho + PAN_TH - 1) / PAN_TH);
    return pan_tables_bytes(B, K) + (((size_t)B * tiles * K * 8 + 255) & ~(size_t)255);
}

int vkn_launch_panoptic_joint(const VknPanopticCfg* c, const float* cls, const float* masks, int B, int N, int ncls,
                              int* panoptic_seg, int* info, int* nseg, int* bbox, void* ws, size_t ws_bytes, hipStream_t st) {
    const int Np = c->num_proposals, T = c->num_thing_classes, Kt = c->max_per_img;
    const int nstuff = N - Np, K = Kt + nstuff;
    if (Np <= 0 || Np > N || T < 0 || Kt <= 0 || Kt > Np * T || nstuff < 0 || T + nstuff > ncls) return VKN_E_ARG;
    if ((size_t)(Np * T + nstuff + K) * 4 > 60 * 1024) return VKN_E_SHAPE;  // selection candidates are ranked in LDS
    if (c->up < 1 || c->Hm <= 0 || c->Wm <= 0 || c->Hb <= 0 || c->Wb <= 0 || c->h <= 0 || c->w <= 0 || c->Ho <= 0 || c->Wo <= 0)
        return VKN_E_ARG;
    if (c->h > c->Hb || c->w > c->Wb) return VKN_E_ARG;  // img_shape is a crop of batch_input_shape
    if (ws_bytes < vkn_panoptic_ws_bytes(B, K, c->Ho, c->Wo)) return VKN_E_WORKSPACE;
    int* wsi = static_cast<int*>(ws);
    int* sel_row = wsi;
    int* sel_label = sel_row + (size_t)B * K;
    int* order = sel_label + (size_t)B * K;
    int* area = order + (size_t)B * K;
    int* orig = area + (size_t)B * K;
    int* seg_of = orig + (size_t)B * K;
    float* sel_score = reinterpret_cast<float*>(seg_of + (size_t)B * K);
    int* err = reinterpret_cast<int*>(sel_score + (size_t)B * K);
    pan_f2* bounds = reinterpret_cast<pan_f2*>(static_cast<char*>(ws) + pan_tables_bytes(B, K));
    if (hipMemsetAsync(area, 0, (size_t)B * K * 2 * sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;
    if (hipMemsetAsync(err, 0, sizeof(int), st) != hipSuccess) return VKN_E_LAUNCH;

    {
        int chunks = (Np * T + 2047) / 2048;  // ~8 candidates per thread
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL(k_pan_select, dim3(chunks, B), dim3(256), (size_t)(Np * T + nstuff) * 4, st, cls, N, ncls, Np, T, Kt, nstuff,
                           sel_row, sel_label, sel_score);
    }
    VKN_CHECK_LAUNCH();

    PanGeom g{};
    g.Hm = c->Hm; g.Wm = c->Wm;
    const int Ha = c->Hm * c->up, Wa = c->Wm * c->up;
    // F.interpolate(scale_factor=up): ATen maps coordinates with 1/scale_factor; size= : with in/out (both in fp32)
    g.lv[0] = PanLevel{c->Hm, c->Wm, (float)(1.0 / (double)c->up), (float)(1.0 / (double)c->up)};
    g.lv[1] = PanLevel{Ha, Wa, (float)Ha / (float)c->Hb, (float)Wa / (float)c->Wb};
    g.lv[2] = PanLevel{c->h, c->w, (float)c->h / (float)c->Ho, (float)c->w / (float)c->Wo};
    g.nlev = (c->h == c->Ho && c->w == c->Wo) ? 2 : 3;
    g.Ho = c->Ho; g.Wo = c->Wo;
    // input-region capacities, from the output tile back
    int ow = PAN_TW, oh = PAN_TH;
    for (int l = g.nlev - 1; l >= 0; --l) {
        g.cap_w[l] = cap_of(ow, g.lv[l].sx, g.lv[l].in_w);
        g.cap_h[l] = cap_of(oh, g.lv[l].sy, g.lv[l].in_h);
        ow = g.cap_w[l]; oh = g.cap_h[l];
    }
    size_t lds = 0;
    for (int l = 0; l < g.nlev; ++l) {
        const int nw = (l == g.nlev - 1) ? PAN_TW : g.cap_w[l + 1], nh = (l == g.nlev - 1) ? PAN_TH : g.cap_h[l + 1];
        lds += (size_t)(3 * nw + 3 * nh) * 4;
        if (l > 0) lds += (size_t)PAN_KB * g.cap_w[l] * g.cap_h[l] * 4;
    }
    lds += (size_t)7 * K * 4;
    // logits footprints of the SURVIVORS of a tile, a chunk of KC at a time (a handful survive for segmentation-like masks; an
    // adversarial input — every kernel everywhere — walks the chunks): 32 footprints keep the workgroup small enough for several per CU
    const size_t ln_bytes = (size_t)g.cap_w[0] * g.cap_h[0] * 4;
    const size_t lds_cap = 150 * 1024;
    if (lds + PAN_KB * ln_bytes > lds_cap) return VKN_E_SHAPE;  // extreme down-scaling: one tile's footprint does not fit LDS
    size_t budget = lds_cap - lds;
    if (budget > 32 * ln_bytes) budget = 32 * ln_bytes;
    int KC = (int)(budget / ln_bytes);
    if (KC >= K) KC = K;
    KC = KC / PAN_KB * PAN_KB;
    lds += (size_t)KC * ln_bytes;
    dim3 grid((c->Wo + PAN_TW - 1) / PAN_TW, (c->Ho + PAN_TH - 1) / PAN_TH, B);
    if ((size_t)2 * PANB_KCH * PANB_TR * g.Wm * 4 > 64 * 1024) return VKN_E_SHAPE;
    // (tried: an XCD-aware unit order that assembles every 128-byte line of `bounds` in one L2 — 168 us against 160: not the limit)
    hipLaunchKernelGGL(k_pan_bounds, dim3((K + PANB_KCH - 1) / PANB_KCH, (grid.y + PANB_TR - 1) / PANB_TR, B), dim3(256),
                       (size_t)2 * PANB_KCH * PANB_TR * g.Wm * 4, st, g, masks, sel_row, K, N, (int)grid.x, (int)grid.y, bounds);
    VKN_CHECK_LAUNCH();
    VKN_ALLOW_FULL_LDS(k_pan_argmax);
    hipLaunchKernelGGL(k_pan_argmax, grid, dim3(PAN_THREADS), lds, st, g, masks, sel_row, sel_score, K, N, panoptic_seg, area, orig, err,
                       vkn_dbg_env("VKN_PAN_NOPRUNE", 0) ? 0 : 1, KC, bounds);  // debug build only: visit all K kernels in every tile
    VKN_CHECK_LAUNCH();

    hipLaunchKernelGGL(k_pan_merge, dim3(B), dim3(64), (size_t)K * 6 * 4, st, sel_row, sel_label, sel_score, order, area, orig, K, T,
                       c->instance_score_thr, c->overlap_thr, seg_of, info, nseg, err, bbox);
    VKN_CHECK_LAUNCH();
    const size_t npx = (size_t)c->Ho * c->Wo;
    size_t blocks = (npx + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pan_relabel, dim3((unsigned)blocks, B), dim3(256), (size_t)K * 16, st, panoptic_seg, seg_of, K, c->Wo, npx, bbox);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
