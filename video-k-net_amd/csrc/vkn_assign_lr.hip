// vkn_assign_lr.hip — the train-time assignment costs of a whole batch straight from the LOW-RES mask logits (round 6).
//
// Reference, per image and stage: MaskHungarianAssigner.assign (knet/det/mask_hungarian_assigner.py:160-274) on
// `F.interpolate(mask_preds, scale_factor=mask_upsample_stride, mode='bilinear', align_corners=False)`
// (knet/det/kernel_update_head.py:122-130 -> knet/det/kernel_iter_head.py:150-156, 225-226): DiceCost (:37-74), MaskCost (:87-113),
// FocalLossCost — the formulas are restated at the top of vkn_assign.hip.
//
// vkn_assign_costs_f32 streams the up-scaled logits three times per image and stage (read x`S`^2 logits -> write the activation
// plane -> read it back into the gather): 0.75 GB per image at 512x1024, 2.5 of the 10.4 ms of a cfg3 training step.  Here the
// up-scaled prediction never exists in memory: a workgroup stages a low-res tile (+ one-pixel halo) of every kernel in LDS, each
// lane interpolates and activates the pixels of ITS kernel in registers — in exactly the layout of the MFMA's B operand — and
// contracts them with the ground truth read once:
//   * workgroup = 4 waves, one per block of 32 kernels (n = 32 wave + lane & 31), all on the same pixels;
//   * one MFMA k-step = 16 consecutive pixels of one up-scaled row: lanes 0..31 hold px 0..7, lanes 32..63 px 8..15 (8 f16 each) —
//     the ground-truth A operand of lane (g = lane & 31, half) is two 16-byte loads of row g, 64 contiguous bytes per row and k-step;
//   * tile = 8 up-scaled rows x 16 S pixels: low-res rows 8 t / S - 1 .. + 8 / S, columns 16 c - 1 .. 16 c + 16 (clamped, as
//     PyTorch's source index is); the horizontal interpolation of a lane's 8 pixels is shared by the tile's 8 rows;
//   * arithmetic: p = clamp(sigmoid(z)) with v_exp_f32 / v_rcp_f32 (1 ulp each), both operands on the two-term f16 split
//     (g_lo p_hi + g_hi p_lo + g_hi p_hi, fp32 accumulate: 2^-22 relative, exact for 0/1 masks), accumulators flushed into a second
//     level (LDS) per tile, row sums of p1^2 and p2 per lane, sum g / sum g^2 of the ground truth riding along (the active waves share
//     the steps), every partial summed in fixed order (fp64) by the finishing kernel — deterministic, independent of the batch size
//     (AL_WGS workgroups per image whatever the batch);
//   * one block of 32 ground truths per pass (blockIdx.z): an image with more of them recomputes the activations per block (a pass
//     over two blocks at once needs 128 accumulator registers — one wave per SIMD — and its instantiation with a `sched_barrier` per
//     step produced wrong sums: docs/LAB_NOTEBOOK.md III.1).
//   The interpolation is PyTorch's expression h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11) with its clamped source indices; where the
//   source coordinate is clamped at the top / left border PyTorch's weights collapse to (1, 0) and ours stay (1 - l, l) on twice the
//   same value — at most one ulp of the logit.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

// Floating-point contraction is OFF in this file: hipcc contracts `a * b + c` in one instantiation of a template and not in another
// (seen here between the one- and two-block forms: an image's costs depended on the widest ground truth of its batch in the last
// bit).  The fused multiply-adds below are written out.
#pragma clang fp contract(off)

#define AL_WGS 128     // workgroups per image (fixed: the order of the partial sums does not depend on the batch)
#define AL_MAXB 16     // images per launch

namespace {

struct AlImg {
    const float* low;   // [N][h][w]
    const float* gt;    // [G][S h][S w]
    const float* cls;   // [N][ncls] or null
    const int* labels;  // [G]
    float* cost;        // [N][G]
    int G;
};
struct AlArgs {
    AlImg im[AL_MAXB];
    float* spart;   // [image][nwg][GBT][2][32][Npad]
    float* rpart;   // [image][nwg][Npad][2]
    float* gpart;   // [image][nwg][GBT][32][2]   (sum g, sum g^2 of the workgroup's pixels)
    int N, Npad, h, w, nwg, tpw, ntiles, GBT;
    float lo1, lo2;
};

// px e of an aligned group of 8 up-scaled pixels: PyTorch's source position (e + 0.5) / S - 0.5 relative to the group's first
// low-res pixel -> left neighbour j0(e) in {-1, 0, ..} and the weight of the right neighbour
template <int S>
__device__ __forceinline__ constexpr int al_j0(int e) { return (2 * e + 1 + S) / (2 * S) - 1; }
template <int S>
__device__ __forceinline__ constexpr float al_l(int e) { return (float)((2 * e + 1 + S) % (2 * S)) / (float)(2 * S); }

__device__ __forceinline__ void al_split8(const float (&p)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        vkn_half2 h2, l2;
        vkn_split_f16x2(p[2 * q], p[2 * q + 1], h2, l2);
        hi[2 * q] = h2[0]; hi[2 * q + 1] = h2[1];
        lo[2 * q] = l2[0]; lo[2 * q + 1] = l2[1];
    }
}

// RAGGED: the map is not a whole number of tiles (w % 16 != 0 or S h % 8 != 0; S w % 8 == 0 still holds, so a lane's 8 pixels are inside
// or outside the map as a whole): steps beyond the right / bottom edge load a clamped (valid) address and are masked out of every sum.
template <int S, bool RAGGED>
__global__ __launch_bounds__(256, 2) void k_assign_lr(const AlArgs A) {
    constexpr int TW = 16 * S, NR = 8 / S + 2, NC = 8 / S + 2, NCL = 18, MS = S, CS = 129;   // (odd stride: the fill's writes and the lanes' reads both walk consecutive banks)
    constexpr int NF = NR * NCL * 128 / 256, NQ = MS * 8;
    extern __shared__ __attribute__((aligned(16))) float al_lds[];   // [NR][NCL][CS], [4 waves][64] ground-truth sums, [4 waves][2][16][64] second-level accumulators
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, li = lane & 31;
    const int b = blockIdx.y, wg = blockIdx.x;
    const int nb = A.Npad >> 5, ngrp = (nb + 3) >> 2;
    const int zn = (int)blockIdx.z % ngrp, zg = (int)blockIdx.z / ngrp;   // group of 128 kernels, block of 32 ground truths
    const bool active = zn * 4 + wave < nb;
    const AlImg& I = A.im[b];
    if (zg * 32 >= I.G) return;   // (uniform: this image has fewer ground truths than the widest of the batch)
    const int H = S * A.h, W = S * A.w;
    const size_t lp = (size_t)A.h * A.w, HP = (size_t)H * W;
    const int tilesx = (W + TW - 1) / TW;
    const float* grow = I.gt + (size_t)min(zg * 32 + li, I.G - 1) * HP + (RAGGED ? 0 : 8 * half);
    // second-level accumulators (the tile sums are flushed into them once per tile): in LDS — 32 registers per lane that would
    // otherwise live across the whole tile loop
    float* TOT = al_lds + NR * NCL * CS + 4 * 64 + wave * (2 * 16 * 64) + lane;
#pragma unroll
    for (int i = 0; i < 2 * 16; ++i) TOT[i * 64] = 0.f;
    float rs1 = 0.f, rs2 = 0.f, gs = 0.f, gq2 = 0.f;
    // sum g / sum g^2 ride along: every active wave holds the same ground truth, active wave w sums the steps q = w (mod active waves)
    const int nact = min(4, nb - zn * 4);
    unsigned gmask = 0;
    for (int q = 0; q < NQ; ++q) gmask |= (unsigned)((q % nact) == wave) << q;
    gmask = (unsigned)__builtin_amdgcn_readfirstlane((int)gmask);
    const float lo1 = A.lo1, lo2 = A.lo2;
    const int t_lo = wg * A.tpw, t_hi = min(A.ntiles, t_lo + A.tpw);
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(I.low), 0, (int)((size_t)A.N * lp * 4), 0x00020000);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int ty = tile / tilesx, tx = tile - ty * tilesx;
        const int r0 = 8 * ty / S - 1, c0 = 16 * tx - 1;
        __syncthreads();   // the previous tile's reads are done
        {
            // the low-res tile: NF values per thread, ALL requested before the first one is stored (a rolled loop is one memory round
            // trip per value), one 32-bit buffer offset per value instead of a 64-bit address.  Rows beyond N: a copy of the last
            // kernel (never read back).  The co-resident workgroup of the CU computes meanwhile.
            float fv[NF];
#pragma unroll
            for (int it = 0; it < NF; ++it) {
                const int i = it * 256 + (int)threadIdx.x;
                const int k = i % NCL, rr = (i / NCL) % NR, nl = i / (NCL * NR);
                const int nn = min(zn * 128 + nl, A.N - 1);
                const int row = min(max(r0 + rr, 0), A.h - 1), col = min(max(c0 + k, 0), A.w - 1);
                fv[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lrs, (nn * (int)lp + row * A.w + col) * 4, 0, 0));
            }
#pragma unroll
            for (int it = 0; it < NF; ++it) {
                const int i = it * 256 + (int)threadIdx.x;
                const int k = i % NCL, rr = (i / NCL) % NR, nl = i / (NCL * NR);
                al_lds[(rr * NCL + k) * CS + nl] = fv[it];
            }
        }
        __syncthreads();
        if (!active) continue;
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc1, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = 0.f;
        float ts1 = 0.f, ts2 = 0.f;
        // the tile's MS x 8 k-steps, flat: step q = 8 m + y = row y of pixel group m.  The ground truth of step q + 1 is requested
        // before step q's arithmetic; the scheduling barrier between steps keeps hipcc from hoisting EVERY step's loads to the top of
        // the tile (256 registers of ground truth, spills)
        const size_t g0 = (size_t)(8 * ty) * W + TW * tx;
        // offset of step q's 8 pixels of this lane (RAGGED: row and column clamped into the map) and whether they exist
        auto goff = [&](int q) -> size_t {
            if (!RAGGED) return g0 + (size_t)(q & 7) * W + 16 * (q >> 3);
            const int yy = min(8 * ty + (q & 7), H - 1), xx = min(TW * tx + 16 * (q >> 3) + 8 * half, W - 8);
            return (size_t)yy * W + xx;
        };
        auto gval = [&](int q) -> float {
            if (!RAGGED) return 1.f;
            return (8 * ty + (q & 7) < H && TW * tx + 16 * (q >> 3) + 8 * half < W) ? 1.f : 0.f;
        };
        f32x4 ga0 = *reinterpret_cast<const f32x4*>(grow + goff(0)), ga1 = *reinterpret_cast<const f32x4*>(grow + goff(0) + 4);
        float hz[NR][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = q >> 3, y = q & 7;
            f32x4 gn0, gn1;
            if (q + 1 < NQ) {
                const size_t gq = goff(q + 1);
                gn0 = *reinterpret_cast<const f32x4*>(grow + gq);
                gn1 = *reinterpret_cast<const f32x4*>(grow + gq + 4);
            }
            if (y == 0) {
                // this lane's 8 pixels: up-scaled columns TW tx + 16 m + 8 half + e; their first low-res column is tile column kb (LDS column kb <-> j0 = -1)
                const int kb = (16 * m + 8 * half) / S;
                float v[NR][NC];
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) v[r][c] = al_lds[(r * NCL + kb + c) * CS + wave * 32 + li];
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        hz[r][e] = __builtin_fmaf(al_l<S>(e), v[r][al_j0<S>(e) + 2], (1.f - al_l<S>(e)) * v[r][al_j0<S>(e) + 1]);
            }
            const float vq = gval(q);
            float p1[8], p2[8];
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float z = __builtin_fmaf(al_l<S>(y), hz[al_j0<S>(y) + 2][e], (1.f - al_l<S>(y)) * hz[al_j0<S>(y) + 1][e]);
                const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
                p1[e] = fmaxf(s, lo1);
                p2[e] = fmaxf(s, lo2);
                if (RAGGED) {
                    q1 = __builtin_fmaf(p1[e], p1[e], q1);
                    q2 += p2[e];
                } else {
                    ts1 = __builtin_fmaf(p1[e], p1[e], ts1);
                    ts2 += p2[e];
                }
            }
            if (RAGGED) {
                ts1 = __builtin_fmaf(vq, q1, ts1);
                ts2 = __builtin_fmaf(vq, q2, ts2);
            }
            half8 h1, l1, h2, l2;
            al_split8(p1, h1, l1);
            al_split8(p2, h2, l2);
            float gv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[e] = RAGGED ? ga0[e] * vq : ga0[e];
                gv[4 + e] = RAGGED ? ga1[e] * vq : ga1[e];
            }
            {   // (predicated, not branched: a branch per step splits the tile into 64 basic blocks and hipcc spills ~500 registers)
                const float gm = ((gmask >> q) & 1u) ? 1.f : 0.f;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s0 += gv[e];
                    s1 = __builtin_fmaf(gv[e], gv[e], s1);
                }
                gs = __builtin_fmaf(gm, s0, gs);
                gq2 = __builtin_fmaf(gm, s1, gq2);
            }
            half8 gh, gl;
            al_split8(gv, gh, gl);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, h1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, l1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, h1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, h2, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, l2, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, h2, acc2, 0, 0, 0);
            if (q + 1 < NQ) {
                ga0 = gn0;
                ga1 = gn1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            TOT[r * 64] += acc1[r];
            TOT[(16 + r) * 64] += acc2[r];
        }
        rs1 += ts1;
        rs2 += ts2;
    }
    // ---- this workgroup's partial sums
    // ground truth: lane (g, half) of wave w holds its share of the steps -> the two halves, then the four waves (fixed order) through LDS
    float* GS = al_lds + NR * NCL * CS;
    gs += __shfl_xor(gs, 32);
    gq2 += __shfl_xor(gq2, 32);
    if (half == 0) {
        GS[(wave * 32 + li) * 2 + 0] = active ? gs : 0.f;
        GS[(wave * 32 + li) * 2 + 1] = active ? gq2 : 0.f;
    }
    __syncthreads();
    if (zn == 0 && threadIdx.x < 64) {
        const int j = threadIdx.x;   // j = 2 g + {0: sum g, 1: sum g^2}
        float* gp = A.gpart + ((((size_t)b * A.nwg + wg) * A.GBT + zg) * 32) * 2;
        gp[j] = (GS[j] + GS[64 + j]) + (GS[128 + j] + GS[192 + j]);
    }
    if (!active) return;
    const int n = (zn * 4 + wave) * 32 + li;
    // S[g][n] (two planes) and the row sums (the pass over the first ground-truth block writes them)
    float* sp = A.spart + ((((size_t)b * A.nwg + wg) * A.GBT + zg) * 2) * 32 * A.Npad;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int g = vkn_cd_row(r, lane);
        sp[(size_t)g * A.Npad + n] = TOT[r * 64];
        sp[(size_t)(32 + g) * A.Npad + n] = TOT[(16 + r) * 64];
    }
    rs1 += __shfl_xor(rs1, 32);
    rs2 += __shfl_xor(rs2, 32);
    if (zg == 0 && half == 0) {
        float* rp = A.rpart + (((size_t)b * A.nwg + wg) * A.Npad + n) * 2;
        rp[0] = rs1;
        rp[1] = rs2;
    }
}

// cost[n][g] of every image: the partial sums of the AL_WGS workgroups meet in fp64, fixed order (the formulas of k_assign_cost,
// vkn_assign.hip).  Block = 64 kernels x 4 quarters of the workgroup range, one ground truth per block: coalesced along n.
__global__ __launch_bounds__(256) void k_assign_cost_lr(const AlArgs A, const VknAssignCfg c, int ncls, double HP) {
    __shared__ double red[6][4][64];
    const int b = blockIdx.z, g = blockIdx.y;
    const AlImg& I = A.im[b];
    if (g >= I.G) return;
    const int ln = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int n = min((int)blockIdx.x * 64 + ln, A.N - 1);
    const int per = (A.nwg + 3) >> 2, k_lo = kq * per, k_hi = min(A.nwg, k_lo + per);
    double s1 = 0.0, s2 = 0.0, sp1sq = 0.0, sp2 = 0.0, sg = 0.0, sgsq = 0.0;
#pragma unroll 8
    for (int k = k_lo; k < k_hi; ++k) {
        const float* sp = A.spart + ((((size_t)b * A.nwg + k) * A.GBT + (g >> 5)) * 2) * 32 * A.Npad;
        const float* rp = A.rpart + (((size_t)b * A.nwg + k) * A.Npad + n) * 2;
        s1 += (double)sp[(size_t)(g & 31) * A.Npad + n];
        s2 += (double)sp[(size_t)(32 + (g & 31)) * A.Npad + n];
        sp1sq += (double)rp[0];
        sp2 += (double)rp[1];
        const float* gp = A.gpart + ((((size_t)b * A.nwg + k) * A.GBT + (g >> 5)) * 32 + (g & 31)) * 2;   // (the same two values for every lane)
        sg += (double)gp[0];
        sgsq += (double)gp[1];
    }
    red[4][kq][ln] = sg; red[5][kq][ln] = sgsq;
    red[0][kq][ln] = s1; red[1][kq][ln] = s2; red[2][kq][ln] = sp1sq; red[3][kq][ln] = sp2;
    __syncthreads();
    if (kq != 0 || (int)blockIdx.x * 64 + ln >= A.N) return;
    s1 = (red[0][0][ln] + red[0][1][ln]) + (red[0][2][ln] + red[0][3][ln]);
    s2 = (red[1][0][ln] + red[1][1][ln]) + (red[1][2][ln] + red[1][3][ln]);
    sp1sq = (red[2][0][ln] + red[2][1][ln]) + (red[2][2][ln] + red[2][3][ln]);
    sp2 = (red[3][0][ln] + red[3][1][ln]) + (red[3][2][ln] + red[3][3][ln]);
    sg = (red[4][0][ln] + red[4][1][ln]) + (red[4][2][ln] + red[4][3][ln]);
    sgsq = (red[5][0][ln] + red[5][1][ln]) + (red[5][2][ln] + red[5][3][ln]);
    double total = 0.0;
    if (c.dice_weight != 0.f) total += (double)c.dice_weight * (-(2.0 * s1) / ((sp1sq + (double)c.dice_eps) + (sgsq + (double)c.dice_eps)));
    if (c.mask_weight != 0.f) {
        const double neg = HP - sp2 - sg + s2;   // sum (1 - p2)(1 - g)
        total += (double)c.mask_weight * (-(s2 + neg) / HP);
    }
    if (c.cls_weight != 0.f && I.cls) {
        const int lab = min(max(I.labels[g], 0), ncls - 1);   // (range-checked by the caller; never read out of bounds)
        const float z = I.cls[(size_t)n * ncls + lab];
        const float p = 1.0f / (1.0f + expf(-z));
        const float negc = -logf(1.f - p + c.focal_eps) * (1.f - c.focal_alpha) * powf(p, c.focal_gamma);
        const float posc = -logf(p + c.focal_eps) * c.focal_alpha * powf(1.f - p, c.focal_gamma);
        total += (double)c.cls_weight * (double)(posc - negc);
    }
    I.cost[(size_t)n * I.G + g] = (float)total;
}

struct AlPlan {
    int Npad, ntiles, tpw, nwg, GBT;
    size_t spart, rpart, gpart, total;
};
// 0 = this shape runs here
int al_plan(int nprob, int N, int Gmax, int h, int w, int S, AlPlan* p) {
    if ((S != 2 && S != 4) || N <= 0 || N > 256 || Gmax <= 0 || Gmax > 256 || h <= 0 || w <= 0 || ((S * w) % 8) != 0) return VKN_E_SHAPE;
    if ((size_t)N * h * w * sizeof(float) >= (1ull << 31)) return VKN_E_SHAPE;   // (the low-res tile is read through 32-bit buffer offsets)
    p->Npad = (N + 31) / 32 * 32;
    p->ntiles = ((S * h + 7) / 8) * ((w + 15) / 16);
    p->tpw = (p->ntiles + AL_WGS - 1) / AL_WGS;
    p->nwg = (p->ntiles + p->tpw - 1) / p->tpw;
    p->GBT = (Gmax + 31) / 32;   // one pass over the image per block of 32 ground truths (the activations are recomputed per pass)
    auto up = [](size_t n) { return (n * sizeof(float) + 255) & ~(size_t)255; };
    p->spart = up((size_t)nprob * p->nwg * p->GBT * 2 * 32 * p->Npad);
    p->rpart = up((size_t)nprob * p->nwg * p->Npad * 2);
    p->gpart = up((size_t)nprob * p->nwg * p->GBT * 32 * 2);
    p->total = p->spart + p->rpart + p->gpart;
    return VKN_OK;
}

}  // namespace

extern "C" {

size_t vkn_assign_lowres_workspace_bytes(int nprob, int N, int Gmax, int h, int w, int S) {
    AlPlan p;
    if (nprob <= 0 || nprob > AL_MAXB || al_plan(nprob, N, Gmax, h, w, S, &p) != VKN_OK) return 0;
    return p.total;
}

int vkn_assign_costs_lowres_batch_f32(const VknAssignCfg* cfg, const VknAssignProblem* probs, int nprob, int N, int ncls, int h, int w,
                                      int S, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !probs || nprob <= 0 || N <= 0 || h <= 0 || w <= 0) return VKN_E_ARG;
    if (nprob > AL_MAXB) return VKN_E_SHAPE;
    int Gmax = 0;
    AlArgs A{};
    for (int b = 0; b < nprob; ++b) {
        const VknAssignProblem& pb = probs[b];
        if (!pb.mask_logits || !pb.gt_masks || !pb.cost_out || pb.G <= 0) return VKN_E_ARG;
        if (cfg->cls_weight != 0.f && pb.cls_logits && (!pb.gt_labels || ncls <= 0)) return VKN_E_ARG;
        if ((reinterpret_cast<uintptr_t>(pb.mask_logits) | reinterpret_cast<uintptr_t>(pb.gt_masks)) & 15) return VKN_E_ALIGN;
        A.im[b] = AlImg{pb.mask_logits, pb.gt_masks, pb.cls_logits, pb.gt_labels, pb.cost_out, pb.G};
        Gmax = pb.G > Gmax ? pb.G : Gmax;
    }
    AlPlan p;
    const int rc = al_plan(nprob, N, Gmax, h, w, S, &p);
    if (rc != VKN_OK) return rc;
    if (!ws || ws_bytes < p.total || (reinterpret_cast<uintptr_t>(ws) & 15)) return VKN_E_WORKSPACE;
    char* base = static_cast<char*>(ws);
    A.spart = reinterpret_cast<float*>(base);
    A.rpart = reinterpret_cast<float*>(base + p.spart);
    A.gpart = reinterpret_cast<float*>(base + p.spart + p.rpart);
    A.N = N; A.Npad = p.Npad; A.h = h; A.w = w; A.nwg = p.nwg; A.tpw = p.tpw; A.ntiles = p.ntiles; A.GBT = p.GBT;
    A.lo1 = cfg->dice_pred_min; A.lo2 = cfg->mask_pred_min;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t HP = (size_t)S * h * S * w;
    const int ngrp = (p.Npad / 32 + 3) / 4;
    const dim3 grid(p.nwg, nprob, ngrp * p.GBT);
    const size_t lds = ((size_t)(8 / S + 2) * 18 * 129 + 4 * 64 + 4 * 2 * 16 * 64) * sizeof(float);
    const bool ragged = (w % 16) != 0 || ((S * h) % 8) != 0;
#define AL_LAUNCH_(SV, RV)                                                              \
    do {                                                                                \
        if (lds > 64 * 1024) VKN_ALLOW_FULL_LDS((k_assign_lr<SV, RV>));                 \
        hipLaunchKernelGGL((k_assign_lr<SV, RV>), grid, dim3(256), lds, st, A);         \
    } while (0)
    if (S == 4) { if (ragged) AL_LAUNCH_(4, true); else AL_LAUNCH_(4, false); }
    else { if (ragged) AL_LAUNCH_(2, true); else AL_LAUNCH_(2, false); }
#undef AL_LAUNCH_
    VKN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_assign_cost_lr, dim3((N + 63) / 64, Gmax, nprob), dim3(256), 0, st, A, *cfg, ncls, (double)HP);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

}  // extern "C"
