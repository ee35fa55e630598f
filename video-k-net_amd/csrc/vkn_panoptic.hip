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
#define PAN_PPT (PAN_TH / 4)  // output pixels per thread: rows fy + 4*j
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

// Workgroup = one 64x16 output tile of one frame (thread = 4 output pixels of one column).
//  1. coefficient tables of every level for this tile, from the output back to the logits;
//  2. the logits footprint of ALL K kernels -> LDS in one burst (one global-latency exposure per tile), and per kernel the
//     footprint's max / min logit.  Bilinear resampling and the sigmoid are monotone convex combinations, so every output value
//     of kernel k in this tile lies in [sigmoid(min_k), sigmoid(max_k)]:  k can neither win a pixel nor reach prob 0.5 here if
//         score_k * sigmoid(max_k) < LB := max_j score_j * sigmoid(min_j)   and   sigmoid(max_k) < 0.5
//     (bounds padded by 1e-5 relative: fp32 rounding of the logits moves a probability by <= 2e-6 relative).  Only the surviving kernels — typically a handful per tile for real
//     segmentation masks — are resampled; the result is identical to visiting all K;
//  3. per batch of PAN_KB survivors: wave kk resamples kernel kk's footprint level by level through LDS; then every thread
//     evaluates the last level at its output pixels, keeps the running arg-max (strict >, ascending k: first maximum wins,
//     as torch.argmax) and counts prob >= 0.5.
__global__ __launch_bounds__(PAN_THREADS) void k_pan_argmax(PanGeom g, const float* __restrict__ masks,
                                                            const int* __restrict__ sel_row,
                                                            const float* __restrict__ sel_score, int K, int N,
                                                            int* __restrict__ ids, int* __restrict__ area,
                                                            int* __restrict__ orig, int* __restrict__ err, int prune, int KC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int rx0[3], ry0[3], rw[3], rh[3];  // input region of each level (absolute origin, extent)
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
    const int ln = g.cap_w[0] * g.cap_h[0], lp = g.cap_w[0];
    const int Lo = take_o(KC * ln);  // word offset of the logits footprints: of every kernel when KC == K, else of one chunk / one batch at a time
    float* const Ls = ldsf + Lo;
    const int b1o = take_o(PAN_KB * g.cap_w[1] * g.cap_h[1]);
    const int b2o = (nl == 3) ? take_o(PAN_KB * g.cap_w[2] * g.cap_h[2]) : 0;

    for (int i = tid; i < K; i += PAN_THREADS) { area_s[i] = 0; orig_s[i] = 0; }

    // ---- 1. coefficient tables, from the output tile back to the logits (once per tile)
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        if (l >= nl) continue;
        const int ox0 = (l == nl - 1) ? X0 : rx0[l + 1], oy0 = (l == nl - 1) ? Y0 : ry0[l + 1];
        const int ow = (l == nl - 1) ? tw : rw[l + 1], oh = (l == nl - 1) ? th : rh[l + 1];
        for (int i = tid; i < ow; i += PAN_THREADS) pan_coef(g.lv[l].sx, ox0 + i, g.lv[l].in_w, tab[l].x0[i], tab[l].x1[i], tab[l].lx[i]);
        for (int i = tid; i < oh; i += PAN_THREADS) pan_coef(g.lv[l].sy, oy0 + i, g.lv[l].in_h, tab[l].y0[i], tab[l].y1[i], tab[l].ly[i]);
        __syncthreads();
        if (tid == 0) {
            rx0[l] = tab[l].x0[0]; rw[l] = tab[l].x1[ow - 1] - tab[l].x0[0] + 1;
            ry0[l] = tab[l].y0[0]; rh[l] = tab[l].y1[oh - 1] - tab[l].y0[0] + 1;
            if (rw[l] + 1 > g.cap_w[l] || rh[l] + 1 > g.cap_h[l]) atomicExch(err, 1);
        }
        __syncthreads();
        if (rw[l] + 1 > g.cap_w[l] || rh[l] + 1 > g.cap_h[l]) return;  // capacity bug: reported through `err`, never silent
        for (int i = tid; i < ow; i += PAN_THREADS) { tab[l].x0[i] -= rx0[l]; tab[l].x1[i] -= rx0[l]; }
        for (int i = tid; i < oh; i += PAN_THREADS) { tab[l].y0[i] -= ry0[l]; tab[l].y1[i] -= ry0[l]; }
        __syncthreads();
    }

    // ---- 2. logits footprint of all K kernels -> LDS, bounds, survivor list
    const int lw = rw[0], lh = rh[0];
    const int lw1 = lw + 1, lwh = lw1 * (lh + 1);  // staged with the replicated column / row
    const float inv_lw = 1.0f / (float)lw1;
    const float* mb = masks + (size_t)b * N * g.Hm * g.Wm;
    const int* rowp = sel_row + (size_t)b * K;
    const float* scp = sel_score + (size_t)b * K;
    const float* lbase = mb + (size_t)ry0[0] * g.Wm + rx0[0];
    const size_t plane = (size_t)g.Hm * g.Wm;
    const bool all_fit = KC >= K;
    // footprint element r of this lane (two per lane cover up to 128 elements; larger footprints loop): source / LDS offsets
    // are per-tile constants, the kernel's plane base is wave-uniform -> per kernel the staging is 2 loads + 2 LDS stores
    if (tid == 0) lb_bits = 0;
    for (int c0 = 0; c0 < K; c0 += KC) {
        const int nc = min(KC, K - c0);
        for (int r0 = 0; r0 < lwh; r0 += 128) {
            int so[2], dof[2];
            bool okr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + lane + 64 * u;
                okr[u] = r < lwh;
                const int rc = min(r, lwh - 1);
                const int yy = (int)(((float)rc + 0.5f) * inv_lw), xx = rc - yy * lw1;
                so[u] = min(yy, lh - 1) * g.Wm + min(xx, lw - 1);
                dof[u] = yy * lp + xx;
            }
            for (int kc0 = wave * 4; kc0 < nc; kc0 += (PAN_THREADS / 64) * 4) {  // 4 kernels x 2 elements in flight per lane
                float v[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kc = min(kc0 + q, nc - 1);
                    const float* pl = lbase + (size_t)rowp[c0 + kc] * plane;
                    v[q][0] = pl[so[0]];
                    v[q][1] = pl[so[1]];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (kc0 + q < nc) {
                        if (okr[0]) Ls[(kc0 + q) * ln + dof[0]] = v[q][0];
                        if (okr[1]) Ls[(kc0 + q) * ln + dof[1]] = v[q][1];
                    }
                }
            }
        }
        __syncthreads();
        // bounds: one thread per kernel walks its footprint (no cross-lane reduction)
        for (int kc = tid; kc < nc; kc += PAN_THREADS) {
            const int k = c0 + kc;
            float mx = -INFINITY, mn = INFINITY;
            for (int yy = 0; yy < lh; ++yy)
                for (int xx = 0; xx < lw; ++xx) {
                    const float v = Ls[kc * ln + yy * lp + xx];
                    mx = fmaxf(mx, v);
                    mn = fminf(mn, v);
                }
            const float s = scp[k];
            const float pm = 1.0f / (1.0f + expf(-mx)), pn = 1.0f / (1.0f + expf(-mn));
            const float hi = s * pm * (1.0f + 1e-5f) + 1e-30f, lo = fmaxf(s * pn * (1.0f - 1e-5f), 0.f);
            khi[k] = hi; kpm[k] = pm * (1.0f + 1e-5f);
            atomicMax(&lb_bits, __float_as_int(lo));  // lo >= 0: the int order of the bits is the float order
        }
        __syncthreads();  // bounds done (and, when chunked, the next chunk may overwrite Ls)
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

    // this thread's output pixels: (fx, fy + 4*j), j < PAN_PPT
    const int fx = tid & 63, fy = tid >> 6;
    const bool okx = fx < tw;
    const PanTab tf = (nl == 3) ? tab[2] : tab[1];
    const int fip = (nl == 3) ? g.cap_w[2] : g.cap_w[1];
    const float flx = tf.lx[okx ? fx : 0];
    bool okp[PAN_PPT];
    int foff[PAN_PPT];   // word offset of the pixel's first tap inside a kernel's last-level buffer
    float fly[PAN_PPT];
    float best[PAN_PPT];
    int bid[PAN_PPT];
#pragma unroll
    for (int j = 0; j < PAN_PPT; ++j) {
        okp[j] = okx && (fy + 4 * j) < th;
        const int yy = okp[j] ? fy + 4 * j : 0;
        foff[j] = tf.y0[yy] * fip + tf.x0[okx ? fx : 0];
        fly[j] = tf.ly[yy];
        best[j] = -INFINITY;
        bid[j] = 0;
    }

    // ---- 3. survivors, PAN_KB at a time: wave kk resamples kernel list[a0 + kk] through the intermediate levels
    for (int a0 = 0; a0 < nact; a0 += PAN_KB) {
        const int nk = min(PAN_KB, nact - a0);
        for (int kq = wave; kq < nk; kq += PAN_THREADS / 64) {
            const int k = list[a0 + kq];
            const int slot = all_fit ? k : kq;
            if (!all_fit) {  // footprints did not all fit: this wave re-stages its kernel's footprint (slot = batch position)
                for (int r = lane; r < lwh; r += 64) {
                    const int yy = (int)(((float)r + 0.5f) * inv_lw), xx = r - yy * lw1;
                    Ls[slot * ln + yy * lp + xx] = lbase[(size_t)rowp[k] * plane + (size_t)min(yy, lh - 1) * g.Wm + min(xx, lw - 1)];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
            {   // level 0: logits footprint -> x up -> sigmoid                                     (rescale_masks :446)
                const int ow = rw[1], oh = rh[1], op = g.cap_w[1], ow1 = ow + 1;
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
                const int ow = rw[2], oh = rh[2], op = g.cap_w[2], ip = g.cap_w[1], ow1 = ow + 1;
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
                const int k = list[a0 + kk];
                const int so = lasto + kk * in;
                const float s = scp[k];
                int c = 0;
#pragma unroll
                for (int j = 0; j < PAN_PPT; ++j) {
                    const float v = pan_lerp(ldsf, so + foff[j], fip, 0, fly[j], 0, flx);
                    const float pj = s * v;
                    if (pj > best[j]) { best[j] = pj; bid[j] = k; }
                    c += __popcll(__ballot(okp[j] && v >= 0.5f));
                }
                if (lane == 0 && c) atomicAdd(&orig_s[k], c);
            }
        }
        __syncthreads();
    }

    int* idp = ids + (size_t)b * g.Ho * g.Wo;
#pragma unroll
    for (int j = 0; j < PAN_PPT; ++j)
        if (okp[j]) { idp[(size_t)(Y0 + fy + 4 * j) * g.Wo + X0 + fx] = bid[j]; atomicAdd(&area_s[bid[j]], 1); }
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

size_t vkn_panoptic_ws_bytes(int B, int K) {
    // sel_row, sel_label, order, area, orig, seg_of (int) + sel_score (float) + err
    return ((size_t)B * K * 7 * 4 + 256 + 255) & ~(size_t)255;
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
    if (ws_bytes < vkn_panoptic_ws_bytes(B, K)) return VKN_E_WORKSPACE;
    int* wsi = static_cast<int*>(ws);
    int* sel_row = wsi;
    int* sel_label = sel_row + (size_t)B * K;
    int* order = sel_label + (size_t)B * K;
    int* area = order + (size_t)B * K;
    int* orig = area + (size_t)B * K;
    int* seg_of = orig + (size_t)B * K;
    float* sel_score = reinterpret_cast<float*>(seg_of + (size_t)B * K);
    int* err = reinterpret_cast<int*>(sel_score + (size_t)B * K);
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
    lds += (size_t)5 * K * 4;
    // logits footprints: all K kernels when they fit (<= 64 KB and what the other buffers leave of 150 KB), else chunks for
    // the bounds pass and one per batch slot for the resampling pass
    const size_t ln_bytes = (size_t)g.cap_w[0] * g.cap_h[0] * 4;
    const size_t lds_cap = 150 * 1024;
    if (lds + PAN_KB * ln_bytes > lds_cap) return VKN_E_SHAPE;  // extreme down-scaling: one tile's footprint does not fit LDS
    size_t budget = lds_cap - lds;
    if (budget > 65536) budget = 65536;
    int KC = (int)(budget / ln_bytes);
    if (KC >= K) KC = K;
    lds += (size_t)KC * ln_bytes;
    VKN_ALLOW_FULL_LDS(k_pan_argmax);
    dim3 grid((c->Wo + PAN_TW - 1) / PAN_TW, (c->Ho + PAN_TH - 1) / PAN_TH, B);
    hipLaunchKernelGGL(k_pan_argmax, grid, dim3(PAN_THREADS), lds, st, g, masks, sel_row, sel_score, K, N, panoptic_seg, area, orig, err,
                       vkn_dbg_env("VKN_PAN_NOPRUNE", 0) ? 0 : 1, KC);  // debug build only: visit all K kernels in every tile
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
