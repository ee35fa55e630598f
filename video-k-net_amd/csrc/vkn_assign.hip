// vkn_assign.hip — train-time one-to-one assignment of kernels to ground-truth masks.
//
// Reference, per image: MaskHungarianAssigner.assign (knet/det/mask_hungarian_assigner.py:160-274) with the shipped costs
// (configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:143-160):
//     cost[n][g] = w_cls  * FocalLossCost(cls_logits)[n][label_g]                       (mmdet 2.18 match cost, restated below)
//                + w_dice * DiceCost  = -2 a / (sum_p p1^2 + eps + sum_p g^2 + eps),   a = sum_p p1 g,  p1 = clamp(sigmoid z, dice_pred_min, 1)   (:37-74)
//                + w_mask * MaskCost  = -(sum_p p2 g + sum_p (1 - p2)(1 - g)) / (H W),                p2 = clamp(sigmoid z, mask_pred_min, 1)   (:87-113)
// then scipy.optimize.linear_sum_assignment on the host and assigned_gt_inds[row] = col + 1                                  (:244-271).
//
// The two [N x P] . [P x G] contractions are the gather kernel with the roles swapped: the left operand is the ground truth
// (rows g) as a REAL-valued operand — the reference down-samples gt masks bilinearly to the assign stride
// (knet/det/knet.py:131) and `hard_target` defaults to False, so their borders are soft; DiceCost / MaskCost use the real
// values (`einsum(pred, target)`, `sum(target * target)`, `1 - target`) —, the streamed operand holds the activated
// predictions as channels — p1 rows in channels [0, Npad), p2 rows in [Npad, 2 Npad) — so ONE launch of k_gather_mfma<., 2>
// yields both sums and sum_p g.  k_assign_act writes the activations (+ fixed-order partial row sums of p1^2 and p2),
// k_assign_gtsq the fixed-order partial sums of g^2, k_assign_cost combines everything in fp64.  (0/1 masks split exactly
// into hi = g, lo = 0: for them the results are those of the binary-operand kernel bit for bit.)
// The LSAP itself is the shortest-augmenting-path algorithm scipy uses, in C++ on the host (vkn_lsap_f32).
#include <hip/hip_runtime.h>
#include <math.h>

#include <limits>
#include <vector>

#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

#define AS_CHUNK 4096  // pixels per workgroup of k_assign_act

// act[0 .. Npad) = p1 rows, act[Npad .. 2 Npad) = p2 rows (rows >= N zero); rowsum[n][chunk][2] = partial (sum p1^2, sum p2).
// two_planes == 0 (lo2 >= lo1: p2 = max(p1, lo2) exactly): only the p1 plane is written — the gather forms the second one on the way
// into LDS (vkn_launch_gather_real: x_alias) — half the bytes of the largest tensor of the assignment.
__global__ __launch_bounds__(256) void k_assign_act(const float* __restrict__ logits, float* __restrict__ act,
                                                    float* __restrict__ rowsum, int N, int Npad, int P, int nchunk, float lo1,
                                                    float lo2, int two_planes) {
    __shared__ float red[2][4];
    const int n = blockIdx.y, ck = blockIdx.x;
    const int p_lo = ck * AS_CHUNK, p_hi = min(P, p_lo + AS_CHUNK);
    float s1 = 0.f, s2 = 0.f;
    float* a1 = act + (size_t)n * P;
    float* a2 = act + (size_t)(Npad + n) * P;
    if (n < N) {
        const float* z = logits + (size_t)n * P;
        for (int p = p_lo + threadIdx.x; p < p_hi; p += 256) {
            const float s = 1.0f / (1.0f + expf(-z[p]));
            const float p1 = fminf(fmaxf(s, lo1), 1.0f), p2 = fminf(fmaxf(s, lo2), 1.0f);
            a1[p] = p1;
            if (two_planes) a2[p] = p2;
            s1 += p1 * p1;
            s2 += p2;
        }
    } else {
        for (int p = p_lo + threadIdx.x; p < p_hi; p += 256) {
            a1[p] = 0.f;
            if (two_planes) a2[p] = 0.f;
        }
    }
    s1 = vkn_wave_sum(s1);
    s2 = vkn_wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0 && n < N) {
        rowsum[((size_t)n * nchunk + ck) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        rowsum[((size_t)n * nchunk + ck) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// gsq[g][chunk] = partial sum_p g^2 (fixed order)
__global__ __launch_bounds__(256) void k_assign_gtsq(const float* __restrict__ gt, float* __restrict__ gsq, int P, int nchunk) {
    __shared__ float red[4];
    const int g = blockIdx.y, ck = blockIdx.x;
    const int p_lo = ck * AS_CHUNK, p_hi = min(P, p_lo + AS_CHUNK);
    const float* z = gt + (size_t)g * P;
    float s = 0.f;
    for (int p = p_lo + threadIdx.x; p < p_hi; p += 256) s += z[p] * z[p];
    s = vkn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) gsq[(size_t)g * nchunk + ck] = (red[0] + red[1]) + (red[2] + red[3]);
}

// S [G][2 Npad] from the gather (S[g][n] = sum p1 g, S[g][Npad + n] = sum p2 g), cnt [G] = sum g -> cost [N][G]
__global__ __launch_bounds__(256) void k_assign_cost(VknAssignCfg c, const float* __restrict__ S1, const float* __restrict__ S2, int lds,
                                                     const float* __restrict__ cnt,
                                                     const float* __restrict__ rowsum, const float* __restrict__ gsq,
                                                     const float* __restrict__ cls, const int* __restrict__ labels, int N,
                                                     int Npad, int G, int ncls, int P, int nchunk, float* __restrict__ cost) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * G) return;
    const int n = i / G, g = i - n * G;
    double sp1sq = 0.0, sp2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {  // fixed order -> deterministic
        sp1sq += (double)rowsum[((size_t)n * nchunk + k) * 2 + 0];
        sp2 += (double)rowsum[((size_t)n * nchunk + k) * 2 + 1];
    }
    const double sg = (double)cnt[g];
    double sgsq = 0.0;
    for (int k = 0; k < nchunk; ++k) sgsq += (double)gsq[(size_t)g * nchunk + k];
    double total = 0.0;
    if (c.dice_weight != 0.f) {
        const double a = (double)S1[(size_t)g * lds + n];
        total += (double)c.dice_weight * (-(2.0 * a) / ((sp1sq + (double)c.dice_eps) + (sgsq + (double)c.dice_eps)));
    }
    if (c.mask_weight != 0.f) {
        const double pos = (double)S2[(size_t)g * lds + n];
        const double neg = (double)P - sp2 - sg + pos;  // sum (1 - p2)(1 - g)
        total += (double)c.mask_weight * (-(pos + neg) / (double)P);
    }
    if (c.cls_weight != 0.f && cls) {
        // mmdet FocalLossCost: p = sigmoid(logit); neg = -log(1 - p + eps) (1 - alpha) p^gamma; pos = -log(p + eps) alpha (1 - p)^gamma
        const int lab = min(max(labels[g], 0), ncls - 1);  // the host wrapper rejects labels outside [0, ncls); never read out of bounds
        const float z = cls[(size_t)n * ncls + lab];
        const float p = 1.0f / (1.0f + expf(-z));
        const float negc = -logf(1.f - p + c.focal_eps) * (1.f - c.focal_alpha) * powf(p, c.focal_gamma);
        const float posc = -logf(p + c.focal_eps) * c.focal_alpha * powf(1.f - p, c.focal_gamma);
        total += (double)c.cls_weight * (double)(posc - negc);
    }
    cost[i] = (float)total;
}

namespace {
struct AssignWs {
    float *act, *rowsum, *gsq, *S, *cnt, *part, *cntp;
};
size_t carve_assign(int N, int G, int P, char* base, AssignWs* w) {
    const size_t Npad = (size_t)(N + 31) / 32 * 32, nchunk = (size_t)(P + AS_CHUNK - 1) / AS_CHUNK;
    const size_t Gg = vkn_gather_groups(1, P), GPT = (size_t)(G + 31) / 32 * 32, C2 = 2 * Npad;
    size_t off = 0;
    auto take = [&](size_t n) {
        float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off = (off + n * sizeof(float) + 255) & ~(size_t)255;
        return r;
    };
    w->act = take(C2 * P);
    w->rowsum = take((size_t)N * nchunk * 2);
    w->gsq = take((size_t)G * nchunk);
    w->S = take((size_t)G * C2);
    w->cnt = take(G);
    w->part = take(Gg * GPT * C2);
    w->cntp = take(Gg * GPT);
    return off;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// Device linear sum assignment: the SAME shortest-augmenting-path algorithm as vkn_lsap_f32 below (scipy's), one WAVEFRONT per
// problem, all state in LDS, fp64 arithmetic — so the train-time assignment needs no device -> host copy of the cost matrix.
// Identical results, including degenerate (tied) matrices, need the identical scan: scipy walks `remaining` (filled in reverse,
// shrunk by swap-removal) position by position and keeps
//      index = it   whenever  spc[j] < lowest  ||  (spc[j] == lowest && row4col[j] == -1).
// In closed form: with m the minimum of spc over the remaining columns and T the positions that reach it, the scan ends at
// max{it in T : column still free} if that set is non-empty, else at min T.  The 64 lanes own the positions it = lane, lane + 64, ..;
// the order-free parts (relaxation of spc / path, minimum, the two position reductions, dual updates) run lane-parallel, the
// order-dependent bookkeeping (swap-removal, augmentation along `path`) on lane 0.  The expression order of every fp64 sum is the
// host function's (no multiplications: nothing to contract).
#define LS_MAXDIM 256
struct VknLsapBatch {
    VknLsapProblem p[VKN_LSAP_MAX_BATCH];
};

__device__ __forceinline__ double ls_wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int ls_wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int ls_wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
#define LS_SYNC()                                  \
    do {                                           \
        __builtin_amdgcn_s_waitcnt(0xc07f);        \
        __builtin_amdgcn_wave_barrier();           \
    } while (0)

__global__ __launch_bounds__(64) void k_lsap(VknLsapBatch batch, int* __restrict__ status) {
    __shared__ double u[LS_MAXDIM], v[LS_MAXDIM], spc[LS_MAXDIM];
    __shared__ int path[LS_MAXDIM], col4row[LS_MAXDIM], row4col[LS_MAXDIM], remaining[LS_MAXDIM];
    __shared__ unsigned char SR[LS_MAXDIM], SC[LS_MAXDIM];
    const VknLsapProblem pb = batch.p[blockIdx.x];
    const int lane = threadIdx.x;
    const int nr = pb.nr, nc = pb.nc;
    const bool transpose = nc < nr;
    const int R = transpose ? nc : nr, Cn = transpose ? nr : nc;  // R <= Cn
    const float* __restrict__ cost = pb.cost;
    const double INF = __builtin_huge_val();
    // element (i, j) of the R x Cn problem
    auto c = [&](int i, int j) -> double { return (double)(transpose ? cost[(size_t)j * nc + i] : cost[(size_t)i * nc + j]); };
    int bad = 0;
    for (int e = lane; e < nr * nc; e += 64) {
        const double x = (double)cost[e];
        if (x != x || x == -INF) bad = 1;  // scipy: "matrix contains invalid numeric entries"
    }
    bad = ls_wave_max_i(bad);
    for (int j = lane; j < Cn; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = lane; i < R; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    LS_SYNC();
    int st = bad ? 1 : 0;
    for (int cur = 0; cur < R && st == 0; ++cur) {
        double minVal = 0.0;
        int num_remaining = Cn;
        for (int j = lane; j < Cn; j += 64) { remaining[j] = Cn - j - 1; SC[j] = 0; spc[j] = INF; }
        for (int i = lane; i < R; i += 64) SR[i] = 0;
        LS_SYNC();
        int sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            double sv[LS_MAXDIM / 64];
            int jv[LS_MAXDIM / 64];
            double lmin = INF;
#pragma unroll
            for (int t = 0; t < LS_MAXDIM / 64; ++t) {
                const int it = lane + 64 * t;
                sv[t] = INF;
                jv[t] = -1;
                if (it < num_remaining) {
                    const int j = remaining[it];
                    const double r = minVal + c(i, j) - ui - v[j];
                    double sj = spc[j];
                    if (r < sj) { path[j] = i; spc[j] = r; sj = r; }
                    sv[t] = sj;
                    jv[t] = j;
                    lmin = fmin(lmin, sj);
                }
            }
            const double lowest = ls_wave_min(lmin);
            if (lowest == INF) { st = 2; break; }  // infeasible
            int first = 0x7fffffff, lastfree = -1;
#pragma unroll
            for (int t = 0; t < LS_MAXDIM / 64; ++t) {
                const int it = lane + 64 * t;
                if (jv[t] >= 0 && sv[t] == lowest) {
                    first = min(first, it);
                    if (row4col[jv[t]] == -1) lastfree = max(lastfree, it);
                }
            }
            first = ls_wave_min_i(first);
            lastfree = ls_wave_max_i(lastfree);
            const int index = lastfree >= 0 ? lastfree : first;
            minVal = lowest;
            const int j = remaining[index];
            const int owner = row4col[j];
            if (owner == -1) sink = j;
            else i = owner;
            LS_SYNC();   // every lane has read remaining[index] / row4col[j] before lane 0 rewrites the list
            --num_remaining;
            if (lane == 0) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining];
            }
            LS_SYNC();
        }
        if (st != 0) break;
        // dual updates (host function: u[cur] += minVal; u[r] += minVal - spc[col4row[r]]; v[j] -= minVal - spc[j])
        for (int r2 = lane; r2 < R; r2 += 64) {
            if (r2 == cur) u[r2] += minVal;
            else if (SR[r2]) u[r2] += minVal - spc[col4row[r2]];
        }
        for (int j = lane; j < Cn; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        LS_SYNC();
        if (lane == 0) {  // augment along the alternating path
            int j = sink;
            while (true) {
                const int r2 = path[j];
                row4col[j] = r2;
                const int t = col4row[r2];
                col4row[r2] = j;
                j = t;
                if (r2 == cur) break;
            }
        }
        LS_SYNC();
    }
    if (lane == 0 && status) status[blockIdx.x] = st;
    // outputs in terms of the ORIGINAL matrix (rows = nr): gt_inds[row] = col + 1 or 0; (row_ind, col_ind) pairs sorted by row
    if (st != 0) {
        // failed (NaN / -inf entries, infeasible): the status word reports it, but the caller reads it asynchronously — one step
        // later — and indexes with these outputs at once.  So they are a VALID dummy assignment (row k <-> column 0 for the
        // min(nr, nc) pairs the caller expects): every downstream gather stays in bounds until the flag is raised.
        const int K = nr < nc ? nr : nc;
        for (int r = lane; r < nr; r += 64)
            if (pb.gt_inds) pb.gt_inds[r] = r < K ? 1 : 0;
        for (int k = lane; k < K; k += 64) {
            if (pb.row_ind) pb.row_ind[k] = k;
            if (pb.col_ind) pb.col_ind[k] = 0;
        }
        return;
    }
    if (!transpose) {
        for (int r = lane; r < nr; r += 64) {
            if (pb.gt_inds) pb.gt_inds[r] = (long long)col4row[r] + 1;
            if (pb.row_ind) pb.row_ind[r] = r;
            if (pb.col_ind) pb.col_ind[r] = col4row[r];
        }
    } else {
        int base = 0;
        for (int j0 = 0; j0 < Cn; j0 += 64) {
            const int j = j0 + lane;
            const int g = j < Cn ? row4col[j] : -1;
            const unsigned long long m = __ballot(g != -1);
            if (j < Cn && pb.gt_inds) pb.gt_inds[j] = (long long)g + 1;
            if (g != -1) {
                const int k = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pb.row_ind) pb.row_ind[k] = j;
                if (pb.col_ind) pb.col_ind[k] = g;
            }
            base += __popcll(m);
        }
    }
}

extern "C" {

size_t vkn_sizeof_assign_cfg(void) { return sizeof(VknAssignCfg); }
size_t vkn_sizeof_lsap_problem(void) { return sizeof(VknLsapProblem); }

size_t vkn_assign_workspace_bytes(int N, int G, int P) {
    if (N <= 0 || G <= 0 || P <= 0) return 0;
    AssignWs w;
    return carve_assign(N, G, P, nullptr, &w);
}

int vkn_assign_costs_f32(const VknAssignCfg* cfg, const float* mask_logits, const float* cls_logits, const float* gt_masks,
                         const int* gt_labels, int N, int G, int ncls, int P, float* cost_out, void* ws, size_t ws_bytes,
                         void* stream) {
    if (!cfg || !mask_logits || !gt_masks || !cost_out || N <= 0 || G <= 0 || P <= 0) return VKN_E_ARG;
    if (cfg->cls_weight != 0.f && cls_logits && (!gt_labels || ncls <= 0)) return VKN_E_ARG;
    const int Npad = (N + 31) / 32 * 32;
    if (Npad > 256 || G > 256) return VKN_E_SHAPE;
    const bool one = 2 * Npad <= 256;   // both activations ride ONE gather launch as 2 Npad channels; more than 128 predictions: two
    if ((reinterpret_cast<uintptr_t>(mask_logits) | reinterpret_cast<uintptr_t>(gt_masks) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return VKN_E_ALIGN;
    AssignWs w;
    if (!ws || ws_bytes < carve_assign(N, G, P, nullptr, &w)) return VKN_E_WORKSPACE;
    carve_assign(N, G, P, static_cast<char*>(ws), &w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunk = (P + AS_CHUNK - 1) / AS_CHUNK;
    // MaskCost's activation is DiceCost's clamped higher (every shipped config: 1e-2 against 1e-3; knet_vis: 0 and 0): one stored plane
    const bool alias = cfg->mask_pred_min >= cfg->dice_pred_min;
    hipLaunchKernelGGL(k_assign_act, dim3(nchunk, Npad), dim3(256), 0, st, mask_logits, w.act, w.rowsum, N, Npad, P, nchunk,
                       cfg->dice_pred_min, cfg->mask_pred_min, alias ? 0 : 1);
    VKN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_assign_gtsq, dim3(nchunk, G), dim3(256), 0, st, gt_masks, w.gsq, P, nchunk);
    VKN_CHECK_LAUNCH();
    // "x" = the activations [1][2 Npad][P], left operand = the (possibly soft) ground truth [G][P]
    const float* S2 = w.S + (one ? (size_t)Npad : (size_t)G * Npad);
    int rc;
    if (one) {
        rc = vkn_launch_gather_real(w.act, gt_masks, w.S, w.cnt, w.part, w.cntp, 1, G, 2 * Npad, P, G, st, alias ? Npad : -1,
                                    cfg->mask_pred_min);
    } else {   // [G][Npad] sums of p1, then of p2 (cnt = sum_p g either time)
        rc = vkn_launch_gather_real(w.act, gt_masks, w.S, w.cnt, w.part, w.cntp, 1, G, Npad, P, G, st);
        if (rc == VKN_OK)
            rc = alias ? vkn_launch_gather_real(w.act, gt_masks, w.S + (size_t)G * Npad, w.cnt, w.part, w.cntp, 1, G, Npad, P, G, st, 0,
                                                cfg->mask_pred_min)
                       : vkn_launch_gather_real(w.act + (size_t)Npad * P, gt_masks, w.S + (size_t)G * Npad, w.cnt, w.part, w.cntp, 1, G,
                                                Npad, P, G, st);
    }
    if (rc != VKN_OK) return rc;
    hipLaunchKernelGGL(k_assign_cost, dim3((N * G + 255) / 256), dim3(256), 0, st, *cfg, w.S, S2, one ? 2 * Npad : Npad, w.cnt,
                       w.rowsum, w.gsq, cls_logits, gt_labels, N, Npad, G, ncls, P, nchunk, cost_out);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

size_t vkn_sizeof_assign_problem(void) { return sizeof(VknAssignProblem); }

int vkn_assign_costs_batch_f32(const VknAssignCfg* cfg, const VknAssignProblem* probs, int nprob, int N, int ncls, int P, void* ws,
                               size_t ws_bytes, void* stream) {
    if (!cfg || !probs || nprob <= 0) return VKN_E_ARG;
    for (int b = 0; b < nprob; ++b) {
        const VknAssignProblem& pb = probs[b];
        const int rc = vkn_assign_costs_f32(cfg, pb.mask_logits, pb.cls_logits, pb.gt_masks, pb.gt_labels, N, pb.G, ncls, P, pb.cost_out,
                                            ws, ws_bytes, stream);
        if (rc != VKN_OK) return rc;
    }
    return VKN_OK;
}

// Rectangular linear sum assignment (minimisation), HOST function: the shortest augmenting path algorithm of
// scipy.optimize.linear_sum_assignment (D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 2016).
// ATTRIBUTION: this function restates scipy/optimize/rectangular_lsap/rectangular_lsap.cpp (SciPy 1.x; Copyright (c) 2019,
// PM Larsen and SciPy developers; BSD 3-Clause License — "Redistribution and use in source and binary forms, with or without
// modification, are permitted provided that the above copyright notice, this list of conditions and the disclaimer are retained";
// the full text is in LICENSES/SCIPY-BSD-3-Clause.txt) closely — same variable roles (u, v, path, col4row, row4col, SR, SC,
// remaining) — because identical tie-breaking requires the identical scan order.  It follows scipy's algorithm including its scan order (`remaining` filled in reverse) and its tie rule
// (prefer a column that is still free), so that degenerate cost matrices resolve the same way.  cost: host fp32 [nr][nc] row-major
// (converted to fp64 as scipy does).  Writes min(nr, nc) pairs sorted by row; returns the number of pairs or a negative error.
int vkn_lsap_f32(const float* cost, int nr, int nc, int* row_ind, int* col_ind) {
    if (!cost || !row_ind || !col_ind || nr < 0 || nc < 0) return VKN_E_ARG;
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;
    const int R = transpose ? nc : nr, Cn = transpose ? nr : nc;  // R <= Cn
    std::vector<double> c((size_t)R * Cn);
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) {
            const double v = (double)cost[(size_t)i * nc + j];
            if (v != v || v == -std::numeric_limits<double>::infinity()) return VKN_E_ARG;  // scipy: "matrix contains invalid numeric entries"
            if (transpose) c[(size_t)j * Cn + i] = v;
            else c[(size_t)i * Cn + j] = v;
        }
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(R, 0.0), v(Cn, 0.0), spc(Cn);
    std::vector<int> path(Cn, -1), col4row(R, -1), row4col(Cn, -1), remaining(Cn);
    std::vector<char> SR(R), SC(Cn);
    for (int cur = 0; cur < R; ++cur) {
        double minVal = 0.0;
        int num_remaining = Cn;
        for (int it = 0; it < Cn; ++it) remaining[it] = Cn - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + c[(size_t)i * Cn + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INF) return VKN_E_ARG;  // infeasible
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r2 = 0; r2 < R; ++r2)
            if (SR[r2] && r2 != cur) u[r2] += minVal - spc[col4row[r2]];
        for (int j = 0; j < Cn; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            const int r2 = path[j];
            row4col[j] = r2;
            const int t = col4row[r2];
            col4row[r2] = j;
            j = t;
            if (r2 == cur) break;
        }
    }
    int n = 0;
    if (!transpose) {
        for (int r2 = 0; r2 < R; ++r2) { row_ind[n] = r2; col_ind[n] = col4row[r2]; ++n; }
    } else {  // rows of the original matrix are the columns here: emit sorted by original row
        for (int j = 0; j < Cn; ++j)
            if (row4col[j] != -1) { row_ind[n] = j; col_ind[n] = row4col[j]; ++n; }
    }
    return n;
}

// One launch, one wavefront per problem (vkn.h).  `probs` is a HOST array; it travels as a kernel argument.
int vkn_lsap_batch_f32(const VknLsapProblem* probs, int nprob, int* status, void* stream) {
    if (!probs || nprob < 0) return VKN_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int b0 = 0; b0 < nprob; b0 += VKN_LSAP_MAX_BATCH) {
        const int nb = (nprob - b0 < VKN_LSAP_MAX_BATCH) ? nprob - b0 : VKN_LSAP_MAX_BATCH;
        VknLsapBatch batch;
        for (int k = 0; k < nb; ++k) {
            const VknLsapProblem& p = probs[b0 + k];
            if (!p.cost || p.nr <= 0 || p.nc <= 0) return VKN_E_ARG;
            if (p.nr > LS_MAXDIM || p.nc > LS_MAXDIM) return VKN_E_SHAPE;
            batch.p[k] = p;
        }
        hipLaunchKernelGGL(k_lsap, dim3(nb), dim3(64), 0, st, batch, status ? status + b0 : nullptr);
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}


}  // extern "C"
