// vkn_train.hip — the BACKWARD building blocks of the [B*N, C] chain (training): with these the chain of a training
// step runs on this library's kernels in both directions instead of torch autograd on library GEMMs.
//   reference ops differentiated here                                        reference
//   nn.Linear (every layer of the chain)                                    knet/kernel_updator.py:36-54, knet/det/kernel_update_head.py:93-134
//   nn.LayerNorm (+ ReLU / sigmoid behind it, + the residual before it)     knet/kernel_updator.py:74-93, knet/det/kernel_update_head.py:206-227
//   nn.MultiheadAttention's softmax(q k^T / sqrt(d)) v                      knet/det/kernel_update_head.py:203-206 (mmcv MultiheadAttention)
// Kernels:
//   k_gemm_tn     dW[n][k] = sum_m dY[m][n] A[m][k], db[n] = sum_m dY[m][n]: the contraction runs over the ROWS of both operands, which is
//                 what v_mfma_f32_32x32x2_f32 wants straight from memory (lane = column, two rows per instruction: both operand loads
//                 are coalesced 128-byte row segments, no LDS staging, exact fp32 products).  A 32 x 32 tile of dW per workgroup of
//                 16 waves = 16 row slices, summed through LDS in fixed order (deterministic).
//   k_ln_fwd      y = act(LayerNorm(in + resid)) per row, (mean, rstd) kept for the backward; a wave per row.
//   k_ln_bwd      dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dz gamma, dz = act'(z) dy (row workgroups) and dgamma = sum dz xhat,
//                 dbeta = sum dz (column workgroups of the same launch; fixed order, no atomics).
//   k_attn_bwd    one workgroup per (frame, head): K, V resident in LDS, a tile of 64 query rows at a time: P recomputed from q, k
//                 (fp32 VALU: exact softmax, the forward's MFMA rounding does not enter), dV += P^T dO, dS = P (dO v^T - D) / sqrt(d),
//                 dK += dS^T q, dQ = dS k; D = dO . O from the saved forward output.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/vkn.h"
#include "vkn_common.h"
#include "vkn_launch.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------ k_gemm_tn
// Every dependent memory round trip of these few-row kernels costs ~1 us (the operands were just written by workgroups on other
// XCDs: nothing is in this XCD's L2), so the kernels below are built to need ONE: all loads of a workgroup are requested before the
// first use.  Here: a 32 x 32 tile of dW per workgroup, its 16 waves are 16 row slices (wave s owns rows 32 u + 2 s, + 1); two
// register sets of 8 row pairs each are requested up front (512 rows: the whole contraction of a 4-frame step), longer
// contractions keep two sets in flight.  The slices are summed through LDS in fixed order (deterministic).
constexpr int TN_THREADS = 1024;
constexpr int TN_U = 8;
constexpr size_t TN_LDS = (size_t)(16 * 32 * 32 + 16 * 32) * sizeof(float);

// 16 waves = QN x QK quadrants (32 x 32 outputs each) x NSL row slices; <1, 1, 16> is the latency form described above (one round
// trip at M <= 512), the only one instantiated.
template <int QN, int QK, int NSL>
__device__ __forceinline__ void tn_tile(const float* __restrict__ Y, int ldy, const float* __restrict__ A, int lda,
                                        float* __restrict__ dW, int ldw, float* __restrict__ db, int M, int Nout, int K, int accumulate,
                                        int bx, int by) {
    static_assert(QN * QK * NSL == 16, "16 waves");
    constexpr int TNN = 32 * QN, TNK = 32 * QK;
    extern __shared__ __attribute__((aligned(16))) float smem_tn[];
    float (*red)[TNN][TNK] = reinterpret_cast<float (*)[TNN][TNK]>(smem_tn);            // [row slice][n][k]: 64 KB
    float (*redb)[TNN] = reinterpret_cast<float (*)[TNN]>(smem_tn + NSL * TNN * TNK);   // [row slice][n]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = wave % (QN * QK), slice = wave / (QN * QK);
    const int qn = quad / QK, qk = quad % QK;
    const int n0 = bx * TNN, k0 = by * TNK;
    const int li = lane & 31, half = lane >> 5;
    const int n = n0 + 32 * qn + li, k = k0 + 32 * qk + li;
    const bool okn = n < Nout, okk = k < K;
    const float* yp = Y + (okn ? n : 0);
    const float* ap = A + (okk ? k : 0);
    const float keepn = okn ? 1.f : 0.f, keepk = okk ? 1.f : 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    float a[TN_U], b[TN_U], an[TN_U], bn[TN_U];
    constexpr int RS = 2 * NSL;      // rows per step over the workgroup
    // rows beyond M are read from row M - 1 (a valid address) and multiplied by zero at use: no divergent branches around the loads
    auto fetch = [&](int mb, float (&fa)[TN_U], float (&fb)[TN_U]) {
#pragma unroll
        for (int u = 0; u < TN_U; ++u) {
            const int mc = min(mb + RS * u + 2 * slice + half, M - 1);
            fa[u] = yp[(size_t)mc * ldy];
            fb[u] = ap[(size_t)mc * lda];
        }
    };
    auto consume = [&](int mb, const float (&fa)[TN_U], const float (&fb)[TN_U]) {
#pragma unroll
        for (int u = 0; u < TN_U; ++u) {
            const float keep = (mb + RS * u + 2 * slice + half) < M ? 1.f : 0.f;
            const float av = fa[u] * (keep * keepn), bv = fb[u] * (keep * keepk);
            asum += av;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    };
    constexpr int SET = RS * TN_U;   // rows per register set
    fetch(0, a, b);
    fetch(SET, an, bn);
    __builtin_amdgcn_sched_barrier(0);
    for (int mb = 0; mb < M; mb += 2 * SET) {
        consume(mb, a, b);
        __builtin_amdgcn_sched_barrier(0);
        // unconditional (clamped addresses; the last two are wasted): a branch here makes the compiler drain every load at the merge
        fetch(mb + 2 * SET, a, b);
        __builtin_amdgcn_sched_barrier(0);
        consume(mb + SET, an, bn);
        __builtin_amdgcn_sched_barrier(0);
        fetch(mb + 3 * SET, an, bn);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[slice][32 * qn + vkn_cd_row(r, lane)][32 * qk + li] = acc[r];
    if (qk == 0) {
        asum += __shfl_xor(asum, 32);
        if (half == 0) redb[slice][32 * qn + li] = asum;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TNN * TNK / TN_THREADS; ++j) {
        const int idx = tid + TN_THREADS * j;
        const int rn = idx / TNK, rk = idx % TNK;
        if (n0 + rn < Nout && k0 + rk < K) {
            float v = 0.f;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) v += red[sl][rn][rk];
            float* o = dW + (size_t)(n0 + rn) * ldw + k0 + rk;
            *o = accumulate ? *o + v : v;
        }
    }
    if (db && by == 0 && tid < TNN && n0 + tid < Nout) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) v += redb[sl][tid];
        db[n0 + tid] = accumulate ? db[n0 + tid] + v : v;
    }
}

__global__ __launch_bounds__(TN_THREADS) void k_gemm_tn(const float* __restrict__ Y, int ldy, const float* __restrict__ A, int lda,
                                                        float* __restrict__ dW, int ldw, float* __restrict__ db, int M, int Nout, int K,
                                                        int accumulate) {
    tn_tile<1, 1, 16>(Y, ldy, A, lda, dW, ldw, db, M, Nout, K, accumulate, blockIdx.x, blockIdx.y);
}

// every weight gradient of a chain's backward in ONE launch: the dW GEMMs are off the critical path (nothing downstream reads them),
// so the host side queues them (chain_train.py) and runs them together when the chain's backward is through
struct TnTab {
    const float* Y[VKN_DW_MAX_ITEMS];
    const float* A[VKN_DW_MAX_ITEMS];
    float* dW[VKN_DW_MAX_ITEMS];
    float* db[VKN_DW_MAX_ITEMS];
    int ldy[VKN_DW_MAX_ITEMS], lda[VKN_DW_MAX_ITEMS], nout[VKN_DW_MAX_ITEMS], k[VKN_DW_MAX_ITEMS];
    int tile0[VKN_DW_MAX_ITEMS + 1];
};

// The batched form stages its operands through LDS instead: a 64 x 128 tile of one dW per workgroup of 4 waves (32 x 64 outputs
// each), the rows in chunks of 32 — float4 global loads, 24 KB per chunk, the next chunk in flight (registers) while the MFMAs of
// this one run.  48 KB of LDS, 2 waves per SIMD: two or three workgroups per CU, so that the matrix pipe has work while a workgroup waits
// for its chunk (one workgroup per CU with 64-row chunks: 67 us for the 412 tiles of a stage with its link — the fetch latency under
// load, 3 us, is twice the 1.7 us of MFMAs it was to hide behind; the register-operand form above with 64 x 128 tiles: 78 us).
// No row slices, so no reduction: the accumulators go straight to global memory.
constexpr int TL_THREADS = 256;
constexpr int TL_MC = 32;                       // rows per chunk
constexpr int TL_BN = 64, TL_BK = 128;
constexpr int TL_CHUNK = TL_MC * (TL_BN + TL_BK);   // floats per chunk: Y part [64][64] then A part [64][128]
constexpr size_t TL_LDS = (size_t)2 * TL_CHUNK * sizeof(float);
constexpr int TL_F4 = TL_CHUNK / 4 / TL_THREADS;    // float4 per thread and chunk (12)

__global__ __launch_bounds__(TL_THREADS, 2) void k_gemm_tn_batch(const TnTab T, int nitems, int M, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smem_tl[];
    const int b = blockIdx.x;
    int it = 0;
    while (it + 1 < nitems && b >= T.tile0[it + 1]) ++it;   // block-uniform
    const int tl = b - T.tile0[it];
    const int K = T.k[it], Nout = T.nout[it];
    const int tk = (K + TL_BK - 1) / TL_BK;
    const int bx = tl / tk, by = tl - bx * tk;
    const float* __restrict__ Y = T.Y[it];
    const float* __restrict__ A = T.A[it];
    const int ldy = T.ldy[it], lda = T.lda[it];
    float* __restrict__ dW = T.dW[it];
    float* __restrict__ db = T.db[it];
    const int n0 = bx * TL_BN, k0 = by * TL_BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qn = wave >> 1, qk = wave & 1;
    const int li = lane & 31, hf = lane >> 5;
    // 16-byte loads where the operand allows them (base, row stride and column count multiples of four floats: uniform per layer);
    // rows / columns outside the matrix are read from a clamped address and multiplied by zero — no branches around the loads
    const bool vy = ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) && ((Nout & 3) == 0);
    const bool va = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((K & 3) == 0);

    f32x4 st[TL_F4];
    // PART(op): float4 slot f of operand `op` ([64][wcols] block): element e = tid + 256 f: row e / (wcols / 4), col 4 (e % (wcols / 4)).
    // fetch = raw loads only (a use right behind a load would wait for it); the keep masks are recomputed when the values go to LDS.
    auto fetch_part = [&](const float* __restrict__ src, int ld, int c0, int lim, bool vec, int m0, int fbeg, int nf, int wcols) {
        if (vec) {
#pragma unroll
            for (int f = 0; f < nf; ++f) {
                const int e = tid + TL_THREADS * f;
                const int m = m0 + e / (wcols / 4), c = c0 + 4 * (e % (wcols / 4));
                st[fbeg + f] = *reinterpret_cast<const f32x4*>(src + (size_t)min(m, M - 1) * ld + (c < lim ? c : 0));
            }
        } else {
#pragma unroll
            for (int f = 0; f < nf; ++f) {
                const int e = tid + TL_THREADS * f;
                const int m = m0 + e / (wcols / 4), c = c0 + 4 * (e % (wcols / 4));
                const float* p = src + (size_t)min(m, M - 1) * ld;
#pragma unroll
                for (int q = 0; q < 4; ++q) st[fbeg + f][q] = p[min(c + q, lim - 1)];
            }
        }
    };
    auto commit_part = [&](float* buf, int c0, int lim, int m0, int fbeg, int nf, int wcols) {
#pragma unroll
        for (int f = 0; f < nf; ++f) {
            const int e = tid + TL_THREADS * f;
            const int m = m0 + e / (wcols / 4), c = c0 + 4 * (e % (wcols / 4));
            f32x4 v = st[fbeg + f];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (m < M && c + q < lim) ? v[q] : 0.f;
            *reinterpret_cast<f32x4*>(buf + 4 * e) = v;
        }
    };
    constexpr int NFY = TL_MC * TL_BN / 4 / TL_THREADS, NFA = TL_MC * TL_BK / 4 / TL_THREADS;
    auto fetch = [&](int m0) {
        fetch_part(Y, ldy, n0, Nout, vy, m0, 0, NFY, TL_BN);
        fetch_part(A, lda, k0, K, va, m0, NFY, NFA, TL_BK);
    };
    auto commit = [&](float* buf, int m0) {
        commit_part(buf, n0, Nout, m0, 0, NFY, TL_BN);
        commit_part(buf + TL_MC * TL_BN, k0, K, m0, NFY, NFA, TL_BK);
    };
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    float asum = 0.f;
    fetch(0);
    commit(smem_tl, 0);
    __syncthreads();
    int cur = 0;
    for (int m0 = 0; m0 < M; m0 += TL_MC) {
        const bool more = m0 + TL_MC < M;
        if (more) fetch(m0 + TL_MC);                   // in flight under the MFMAs below
        const float* yb = smem_tl + cur * TL_CHUNK + 32 * qn + li;
        const float* ab = smem_tl + cur * TL_CHUNK + TL_MC * TL_BN + 64 * qk + li;
        // operands of 8 steps per group, the next group's LDS reads issued before this group's MFMAs
        constexpr int TG = 8;
        float oa[2][TG], ob0[2][TG], ob1[2][TG];
        auto lread = [&](int g, float (&xa)[TG], float (&x0)[TG], float (&x1)[TG]) {
#pragma unroll
            for (int u = 0; u < TG; ++u) {
                const int rr = 2 * (g * TG + u) + hf;
                xa[u] = yb[rr * TL_BN];
                x0[u] = ab[rr * TL_BK];
                x1[u] = ab[rr * TL_BK + 32];
            }
        };
        lread(0, oa[0], ob0[0], ob1[0]);
#pragma unroll
        for (int g = 0; g < TL_MC / 2 / TG; ++g) {
            if (g + 1 < TL_MC / 2 / TG) lread(g + 1, oa[(g + 1) & 1], ob0[(g + 1) & 1], ob1[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);    // the reads stay ahead of this group's MFMAs
#pragma unroll
            for (int u = 0; u < TG; ++u) {
                asum += oa[g & 1][u];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[g & 1][u], ob0[g & 1][u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[g & 1][u], ob1[g & 1][u], acc[1], 0, 0, 0);
            }
        }
        if (more) commit(smem_tl + (cur ^ 1) * TL_CHUNK, m0 + TL_MC);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + 32 * qn + vkn_cd_row(r, lane), k = k0 + 64 * qk + 32 * blk + li;
            if (n < Nout && k < K) dW[(size_t)n * K + k] = acc[blk][r];
        }
    if (db && by == 0 && qk == 0) {
        asum += __shfl_xor(asum, 32);
        const int n = n0 + 32 * qn + li;
        if (hf == 0 && n < Nout) db[n] = asum;
    }
}

// ------------------------------------------------------------------------------------------------------------------ k_split_batch
// The bf16x3 tile images of MANY weights in one launch (both orientations of every Linear weight of a stage: the weights change every
// training step).  Image layout as k_split_w3 (vkn_update.hip): [ceil(Nout/256)][K/32][plane 3][q 4][row 256][8] bf16.  A workgroup writes
// one 48-KB tile: thread = row, 32 k values -> twelve 16-byte stores, consecutive threads consecutive 16 bytes.
struct SplitTab {
    const float* W[VKN_SPLIT_MAX_ITEMS];
    __bf16* dst[VKN_SPLIT_MAX_ITEMS];
    long long ldn[VKN_SPLIT_MAX_ITEMS], ldk[VKN_SPLIT_MAX_ITEMS];
    int nout[VKN_SPLIT_MAX_ITEMS], k[VKN_SPLIT_MAX_ITEMS], kvalid[VKN_SPLIT_MAX_ITEMS];
    int tile0[VKN_SPLIT_MAX_ITEMS + 1];
};

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 bh = (__bf16)v;
    const float r1 = v - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const __bf16 bl = (__bf16)(r1 - (float)bm);
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}

__global__ __launch_bounds__(256) void k_split_batch(const SplitTab T, int nitems) {
    const int b = blockIdx.x;
    int it = 0;
    while (it + 1 < nitems && b >= T.tile0[it + 1]) ++it;   // block-uniform
    const int tl = b - T.tile0[it];
    const int K = T.k[it], Nout = T.nout[it], kvalid = T.kvalid[it];
    const int ktiles = K >> 5;
    const int nt = tl / ktiles, kt = tl - nt * ktiles;
    const float* __restrict__ W = T.W[it];
    const long long ldn = T.ldn[it], ldk = T.ldk[it];
    const int row = threadIdx.x, n = nt * 256 + row;
    float v[32];
    if (n < Nout) {
        if (ldk == 1 && kt * 32 + 32 <= kvalid && (ldn & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
            const f32x4* src = reinterpret_cast<const f32x4*>(W + (size_t)n * ldn + kt * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 t = src[j];
                v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const int k = kt * 32 + kk;
                v[kk] = k < kvalid ? W[(size_t)n * ldn + (size_t)k * ldk] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) v[kk] = 0.f;
    }
    u16x8* dst = reinterpret_cast<u16x8*>(T.dst[it] + ((size_t)nt * ktiles + kt) * (3 * 4 * 256 * 8));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned short a, c, d;
            split3(v[8 * q + e], a, c, d);
            h[e] = a; m[e] = c; l[e] = d;
        }
        dst[(0 * 4 + q) * 256 + row] = h;
        dst[(1 * 4 + q) * 256 + row] = m;
        dst[(2 * 4 + q) * 256 + row] = l;
    }
}

// ------------------------------------------------------------------------------------------------------------------ LayerNorm
constexpr int LN_MAXV = 4;   // C <= 256: up to four elements per lane

__device__ __forceinline__ float ln_act(float z, int act) {
    if (act == 1) return z > 0.f ? z : 0.f;
    if (act == 2) return 1.f / (1.f + expf(-z));
    return z;
}

// grid = ceil(M / 4), 256 threads: a wave per row
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ in, int ldi, const float* __restrict__ resid, int ldr,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                float* __restrict__ out, int ldo, float* __restrict__ stats, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float x[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane + 64 * j;
        x[j] = 0.f;
        if (c < C) {
            x[j] = in[(size_t)row * ldi + c];
            if (resid) x[j] += resid[(size_t)row * ldr + c];
        }
        s += x[j];
    }
    const float mean = vkn_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const float dlt = (lane + 64 * j < C) ? x[j] - mean : 0.f;
        q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(vkn_wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
            const float z = (x[j] - mean) * rstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
            out[(size_t)row * ldo + c] = ln_act(z, act);
        }
    }
    if (stats && lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

constexpr int LNB_ROWS = 16;      // rows per row-block of k_ln_bwd: a wave per row
constexpr int LNB_THREADS = 1024;

__device__ __forceinline__ float ln_dz(float dy, float z, int act) {
    if (act == 1) return z > 0.f ? dy : 0.f;
    if (act == 2) {
        const float sg = 1.f / (1.f + expf(-z));
        return dy * sg * (1.f - sg);
    }
    return dy;
}

// ONE launch, two kinds of workgroups, each with a single memory round trip:
//   blocks [0, nrb)        16 rows each, a wave per row: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dz gamma
//   blocks [nrb, nrb + ncb) 32 columns each, 32 row slices: dgamma[c] = sum_m dz xhat, dbeta[c] = sum_m dz (recomputed from the same
//                          inputs — cheaper than a partial buffer and a second, dependent launch), fixed summation order
__global__ __launch_bounds__(LNB_THREADS) void k_ln_bwd(const float* __restrict__ dy, int lddy, const float* __restrict__ in, int ldi,
                                                        const float* __restrict__ resid, int ldr, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ stats, int act,
                                                        float* __restrict__ dx, int lddx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int M, int C, int nrb) {
    __shared__ float red[2][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x < nrb) {
        const int row = blockIdx.x * LNB_ROWS + (tid >> 6);
        if (row >= M) return;
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float xv[LN_MAXV], dv[LN_MAXV], gam[LN_MAXV], bet[LN_MAXV];
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            const bool ok = c < C;
            const int cc = ok ? c : 0;
            xv[j] = in[(size_t)row * ldi + cc];
            if (resid) xv[j] += resid[(size_t)row * ldr + cc];
            dv[j] = dy[(size_t)row * lddy + cc];
            gam[j] = gamma ? gamma[cc] : 1.f;
            bet[j] = beta ? beta[cc] : 0.f;
        }
        float xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const bool ok = lane + 64 * j < C;
            xh[j] = (xv[j] - mean) * rstd;
            const float dz = ln_dz(dv[j], xh[j] * gam[j] + bet[j], act);
            g[j] = ok ? dz * gam[j] : 0.f;
            s1 += g[j];
            s2 += g[j] * xh[j];
        }
        const float invC = 1.0f / (float)C;
        s1 = vkn_wave_sum(s1) * invC;
        s2 = vkn_wave_sum(s2) * invC;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < C) dx[(size_t)row * lddx + c] = rstd * (g[j] - s1 - xh[j] * s2);
        }
        return;
    }
    // ---- column gradients
    const int cb = blockIdx.x - nrb;
    const int cl = tid & 31, sl = tid >> 5;          // column of the block, row slice
    const int c = cb * 32 + cl;
    const bool okc = c < C;
    const int cc = okc ? c : 0;
    const float gam = gamma ? gamma[cc] : 1.f, bet = beta ? beta[cc] : 0.f;
    float pg = 0.f, pb = 0.f;
    constexpr int CU = 8;                             // rows per step: all their loads are requested together
    for (int m0 = sl; m0 < M; m0 += 32 * CU) {
        float xv[CU], dv[CU], mu[CU], rs[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int m = min(m0 + 32 * u, M - 1);
            mu[u] = stats[2 * m];
            rs[u] = stats[2 * m + 1];
            xv[u] = in[(size_t)m * ldi + cc];
            if (resid) xv[u] += resid[(size_t)m * ldr + cc];
            dv[u] = dy[(size_t)m * lddy + cc];
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const float xh = (xv[u] - mu[u]) * rs[u];
            float dz = ln_dz(dv[u], xh * gam + bet, act);
            dz = (m0 + 32 * u < M) ? dz : 0.f;
            pg += dz * xh;
            pb += dz;
        }
    }
    red[0][sl][cl] = pg;
    red[1][sl][cl] = pb;
    __syncthreads();
    if (tid < 64) {
        const int which = tid >> 5;
        float v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) v += red[which][s2][cl];
        float* o = which ? dbeta : dgamma;
        if (o && okc) o[c] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------ gated update
// The element-wise core of `KernelUpdator.forward` (knet/kernel_updator.py:70-90) between its GEMMs, as three kernels in each direction:
//   gate product    G = param_in * input_in                                   (the first halves of the packed [M][2C] layer outputs)
//   mix             F = sigmoid(LN_norm_in(UG)) LN_norm_out(param_out) + sigmoid(LN_input_norm_in(IG)) LN_input_norm_out(input_out)
// with GT = [IG | UG] the packed output of the two gate layers.  Everything row-local: a wave per row, four LayerNorms = eight wave
// reductions; the backward's eight parameter-gradient vectors are column sums computed by extra workgroups of the same launch.
struct UpdNorms {   // gamma / beta of norm_in (update gate), norm_out (param_out), input_norm_in (input gate), input_norm_out (input_out)
    const float *in_w, *in_b, *out_w, *out_b, *iin_w, *iin_b, *iout_w, *iout_b;
    const float *ig_b, *ug_b;   // biases of input_gate / update_gate (or null), added to the packed gate GEMM output here
};
struct UpdNormGrads {
    float *in_w, *in_b, *out_w, *out_b, *iin_w, *iin_b, *iout_w, *iout_b;
};

__global__ __launch_bounds__(256) void k_gprod_fwd(const float* __restrict__ P, const float* __restrict__ I, int ld, float* __restrict__ G,
                                                   int M, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;     // one float4
    const int c4 = C >> 2;
    if (idx >= M * c4) return;
    const int m = idx / c4, c = (idx - m * c4) * 4;
    const f32x4 p = *reinterpret_cast<const f32x4*>(P + (size_t)m * ld + c), i = *reinterpret_cast<const f32x4*>(I + (size_t)m * ld + c);
    *reinterpret_cast<f32x4*>(G + (size_t)m * C + c) = p * i;
}

// dP[:, :C] = dG * I[:, :C], dI[:, :C] = dG * P[:, :C]  (the second halves are written by k_mix_bwd)
__global__ __launch_bounds__(256) void k_gprod_bwd(const float* __restrict__ dG, const float* __restrict__ P, const float* __restrict__ I,
                                                   int ld, float* __restrict__ dP, float* __restrict__ dI, int M, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int c4 = C >> 2;
    if (idx >= M * c4) return;
    const int m = idx / c4, c = (idx - m * c4) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dG + (size_t)m * C + c);
    const f32x4 p = *reinterpret_cast<const f32x4*>(P + (size_t)m * ld + c), i = *reinterpret_cast<const f32x4*>(I + (size_t)m * ld + c);
    *reinterpret_cast<f32x4*>(dP + (size_t)m * ld + c) = g * i;
    *reinterpret_cast<f32x4*>(dI + (size_t)m * ld + c) = g * p;
}

__device__ __forceinline__ void ln_stats(const float (&x)[LN_MAXV], int lane, int C, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) s += (lane + 64 * j < C) ? x[j] : 0.f;
    mean = vkn_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const float d = (lane + 64 * j < C) ? x[j] - mean : 0.f;
        q += d * d;
    }
    rstd = 1.0f / sqrtf(vkn_wave_sum(q) / (float)C + eps);
}

// stats [M][8] = (mean, rstd) of norm_in(UG), norm_out(param_out), input_norm_in(IG), input_norm_out(input_out)
__global__ __launch_bounds__(256) void k_mix_fwd(const float* __restrict__ GT, const float* __restrict__ P, const float* __restrict__ I,
                                                 int ld, UpdNorms nw, float eps, float* __restrict__ F, float* __restrict__ stats, int M,
                                                 int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float ig[LN_MAXV], ug[LN_MAXV], po[LN_MAXV], io[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = min(lane + 64 * j, C - 1);
        ig[j] = GT[(size_t)row * ld + c] + (nw.ig_b ? nw.ig_b[c] : 0.f);
        ug[j] = GT[(size_t)row * ld + C + c] + (nw.ug_b ? nw.ug_b[c] : 0.f);
        po[j] = P[(size_t)row * ld + C + c];
        io[j] = I[(size_t)row * ld + C + c];
    }
    float mu[4], rs[4];
    ln_stats(ug, lane, C, eps, mu[0], rs[0]);
    ln_stats(po, lane, C, eps, mu[1], rs[1]);
    ln_stats(ig, lane, C, eps, mu[2], rs[2]);
    ln_stats(io, lane, C, eps, mu[3], rs[3]);
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
            const float u = 1.f / (1.f + expf(-((ug[j] - mu[0]) * rs[0] * nw.in_w[c] + nw.in_b[c])));
            const float o = (po[j] - mu[1]) * rs[1] * nw.out_w[c] + nw.out_b[c];
            const float g = 1.f / (1.f + expf(-((ig[j] - mu[2]) * rs[2] * nw.iin_w[c] + nw.iin_b[c])));
            const float q = (io[j] - mu[3]) * rs[3] * nw.iout_w[c] + nw.iout_b[c];
            F[(size_t)row * C + c] = u * o + g * q;
        }
    }
    if (lane < 4) {
        stats[(size_t)row * 8 + 2 * lane] = mu[lane];
        stats[(size_t)row * 8 + 2 * lane + 1] = rs[lane];
    }
}

// per element: the four normalised values, the two gates, and the four dz (gradients at the LayerNorm outputs)
struct MixElem {
    float xh[4], dz[4];
};
__device__ __forceinline__ MixElem mix_elem(float dF, float ugr, float por, float igr, float ior, const float* st, const float (&w)[4],
                                            const float (&b)[4]) {
    MixElem e;
    e.xh[0] = (ugr - st[0]) * st[1];
    e.xh[1] = (por - st[2]) * st[3];
    e.xh[2] = (igr - st[4]) * st[5];
    e.xh[3] = (ior - st[6]) * st[7];
    const float u = 1.f / (1.f + expf(-(e.xh[0] * w[0] + b[0])));
    const float o = e.xh[1] * w[1] + b[1];
    const float g = 1.f / (1.f + expf(-(e.xh[2] * w[2] + b[2])));
    const float q = e.xh[3] * w[3] + b[3];
    e.dz[0] = dF * o * u * (1.f - u);
    e.dz[1] = dF * u;
    e.dz[2] = dF * q * g * (1.f - g);
    e.dz[3] = dF * g;
    return e;
}

// blocks [0, nrb): 16 rows each (a wave per row) -> dGT [M][2C] = [dIG | dUG], dP[:, C:], dI[:, C:];  blocks [nrb, ..): 32 columns each ->
// the eight parameter gradients (fixed summation order)
__global__ __launch_bounds__(LNB_THREADS) void k_mix_bwd(const float* __restrict__ dF, const float* __restrict__ GT,
                                                         const float* __restrict__ P, const float* __restrict__ I, int ld, UpdNorms nw,
                                                         const float* __restrict__ stats, float* __restrict__ dGT, float* __restrict__ dP,
                                                         float* __restrict__ dI, UpdNormGrads gw, int M, int C, int nrb) {
    __shared__ float red[8][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x < nrb) {
        const int row = blockIdx.x * LNB_ROWS + (tid >> 6);
        if (row >= M) return;
        float st[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) st[k] = stats[(size_t)row * 8 + k];
        float g[4][LN_MAXV], xh[4][LN_MAXV];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            const bool ok = c < C;
            const int cc = ok ? c : 0;
            const float w[4] = {nw.in_w[cc], nw.out_w[cc], nw.iin_w[cc], nw.iout_w[cc]};
            const float b[4] = {nw.in_b[cc], nw.out_b[cc], nw.iin_b[cc], nw.iout_b[cc]};
            const MixElem e = mix_elem(dF[(size_t)row * C + cc], GT[(size_t)row * ld + C + cc] + (nw.ug_b ? nw.ug_b[cc] : 0.f),
                                       P[(size_t)row * ld + C + cc], GT[(size_t)row * ld + cc] + (nw.ig_b ? nw.ig_b[cc] : 0.f),
                                       I[(size_t)row * ld + C + cc], st, w, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xh[k][j] = e.xh[k];
                g[k][j] = ok ? e.dz[k] * w[k] : 0.f;
                s1[k] += g[k][j];
                s2[k] += g[k][j] * e.xh[k];
            }
        }
        const float invC = 1.0f / (float)C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[k] = vkn_wave_sum(s1[k]) * invC;
            s2[k] = vkn_wave_sum(s2[k]) * invC;
        }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < C) {
                dGT[(size_t)row * ld + C + c] = st[1] * (g[0][j] - s1[0] - xh[0][j] * s2[0]);   // d UG
                dP[(size_t)row * ld + C + c] = st[3] * (g[1][j] - s1[1] - xh[1][j] * s2[1]);    // d param_out
                dGT[(size_t)row * ld + c] = st[5] * (g[2][j] - s1[2] - xh[2][j] * s2[2]);       // d IG
                dI[(size_t)row * ld + C + c] = st[7] * (g[3][j] - s1[3] - xh[3][j] * s2[3]);    // d input_out
            }
        }
        return;
    }
    // ---- column gradients
    const int cb = blockIdx.x - nrb;
    const int cl = tid & 31, sl = tid >> 5;
    const int c = cb * 32 + cl;
    const bool okc = c < C;
    const int cc = okc ? c : 0;
    const float w[4] = {nw.in_w[cc], nw.out_w[cc], nw.iin_w[cc], nw.iout_w[cc]};
    const float b[4] = {nw.in_b[cc], nw.out_b[cc], nw.iin_b[cc], nw.iout_b[cc]};
    const float gbi = nw.ig_b ? nw.ig_b[cc] : 0.f, gbu = nw.ug_b ? nw.ug_b[cc] : 0.f;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int CU = 4;
    for (int m0 = sl; m0 < M; m0 += 32 * CU) {
        float vF[CU], vu[CU], vo[CU], vi[CU], vq[CU], st[CU][8];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int m = min(m0 + 32 * u, M - 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) st[u][k] = stats[(size_t)m * 8 + k];
            vF[u] = dF[(size_t)m * C + cc];
            vu[u] = GT[(size_t)m * ld + C + cc] + gbu;
            vo[u] = P[(size_t)m * ld + C + cc];
            vi[u] = GT[(size_t)m * ld + cc] + gbi;
            vq[u] = I[(size_t)m * ld + C + cc];
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const MixElem e = mix_elem(vF[u], vu[u], vo[u], vi[u], vq[u], st[u], w, b);
            const float keep = (m0 + 32 * u < M) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pg[k] += keep * e.dz[k] * e.xh[k];
                pb[k] += keep * e.dz[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[2 * k][sl][cl] = pg[k];
        red[2 * k + 1][sl][cl] = pb[k];
    }
    __syncthreads();
    if (tid < 256) {
        const int which = tid >> 5;
        float v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) v += red[which][s2][cl];
        float* outs[8] = {gw.in_w, gw.in_b, gw.out_w, gw.out_b, gw.iin_w, gw.iin_b, gw.iout_w, gw.iout_b};
        float* o = outs[which];
        if (o && okc) o[c] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------ attention
constexpr int AB_THREADS = 512;

// grid = (heads, B).  LDS: Ks, Vs [Nk][HD+1]; Qt, dOt [TQ][HD+1]; Pt [TQ][Nk+1]; Dt [TQ]
template <int HD, int TQ>
__global__ __launch_bounds__(AB_THREADS) void k_attn_bwd(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                         const float* __restrict__ Vp, int ldkv, const float* __restrict__ O, int ldo,
                                                         const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq,
                                                         float* __restrict__ dK, float* __restrict__ dV, int lddkv, int Nq, int Nk,
                                                         float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem_ab[];
    constexpr int LD = HD + 1;
    float* Ks = smem_ab;
    float* Vs = Ks + (size_t)Nk * LD;
    float* Qt = Vs + (size_t)Nk * LD;
    float* dOt = Qt + (size_t)TQ * LD;
    float* Pt = dOt + (size_t)TQ * LD;
    float* Dt = Pt + (size_t)TQ * (Nk + 1);
    const int LP = Nk + 1;
    const int tid = threadIdx.x;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t qrow0 = (size_t)b * Nq, krow0 = (size_t)b * Nk;
    const int col0 = h * HD;

    for (int idx = tid; idx < Nk * HD; idx += AB_THREADS) {
        const int j = idx / HD, d = idx - j * HD;
        Ks[j * LD + d] = Kp[(krow0 + j) * ldkv + col0 + d];
        Vs[j * LD + d] = Vp[(krow0 + j) * ldkv + col0 + d];
    }
    // (j, half of the head width) accumulators of dK / dV
    constexpr int HH = HD / 2;
    const int aj = tid >> 1, ah = tid & 1;
    float dk[HH], dv[HH];
#pragma unroll
    for (int d = 0; d < HH; ++d) dk[d] = dv[d] = 0.f;
    // (query row of the tile, 1 of NP column parts)
    constexpr int NP = AB_THREADS / TQ;      // 8 / 16 / 32 parts for TQ = 64 / 32 / 16
    constexpr int PER = HD / NP > 0 ? HD / NP : 1;   // head-width columns of dQ per part (parts beyond the head width idle)
    const int ti = tid / NP, tp = tid - ti * NP;

    for (int i0 = 0; i0 < Nq; i0 += TQ) {
        __syncthreads();   // the previous tile's readers are done (first round: K / V staged)
        for (int idx = tid; idx < TQ * HD; idx += AB_THREADS) {
            const int i = idx / HD, d = idx - i * HD;
            const bool ok = i0 + i < Nq;
            Qt[i * LD + d] = ok ? Q[(qrow0 + i0 + i) * ldq + col0 + d] : 0.f;
            dOt[i * LD + d] = ok ? dO[(qrow0 + i0 + i) * lddo + col0 + d] : 0.f;
        }
        if (tid < TQ) {
            float s = 0.f;
            if (i0 + tid < Nq)
                for (int d = 0; d < HD; ++d) s += dO[(qrow0 + i0 + tid) * lddo + col0 + d] * O[(qrow0 + i0 + tid) * ldo + col0 + d];
            Dt[tid] = s;
        }
        __syncthreads();
        // ---- scores, softmax: P into Pt
        {
            const bool ok = i0 + ti < Nq;
            float q[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) q[d] = Qt[ti * LD + d];
            float mx = -INFINITY;
            for (int j = tp; j < Nk; j += NP) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) s += q[d] * Ks[j * LD + d];
                s *= scale;
                Pt[ti * LP + j] = s;
                mx = fmaxf(mx, s);
            }
            for (int o = 1; o < NP; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
            for (int j = tp; j < Nk; j += NP) {
                const float e = expf(Pt[ti * LP + j] - mx);
                Pt[ti * LP + j] = e;
                sum += e;
            }
            for (int o = 1; o < NP; o <<= 1) sum += __shfl_xor(sum, o);
            const float inv = ok ? 1.0f / sum : 0.f;   // rows beyond Nq contribute nothing
            for (int j = tp; j < Nk; j += NP) Pt[ti * LP + j] *= inv;
        }
        __syncthreads();
        // ---- dV += P^T dO
        if (aj < Nk) {
            for (int i = 0; i < TQ; ++i) {
                const float p = Pt[i * LP + aj];
#pragma unroll
                for (int d = 0; d < HH; ++d) dv[d] += p * dOt[i * LD + ah * HH + d];
            }
        }
        __syncthreads();
        // ---- dS = P (dO v^T - D) scale, in place
        {
            float g[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) g[d] = dOt[ti * LD + d];
            const float Di = Dt[ti];
            for (int j = tp; j < Nk; j += NP) {
                float dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) dp += g[d] * Vs[j * LD + d];
                Pt[ti * LP + j] = Pt[ti * LP + j] * (dp - Di) * scale;
            }
        }
        __syncthreads();
        // ---- dK += dS^T q ; dQ = dS k
        if (aj < Nk) {
            for (int i = 0; i < TQ; ++i) {
                const float s = Pt[i * LP + aj];
#pragma unroll
                for (int d = 0; d < HH; ++d) dk[d] += s * Qt[i * LD + ah * HH + d];
            }
        }
        if (i0 + ti < Nq) {
            // the NP parts of a row split the head width; with NP > HD the surplus parts idle
            const int d0 = tp * PER;
            if (d0 < HD) {
                float accq[PER];
#pragma unroll
                for (int d = 0; d < PER; ++d) accq[d] = 0.f;
                for (int j = 0; j < Nk; ++j) {
                    const float s = Pt[ti * LP + j];
#pragma unroll
                    for (int d = 0; d < PER; ++d) accq[d] += s * Ks[j * LD + d0 + d];
                }
#pragma unroll
                for (int d = 0; d < PER; ++d) dQ[(qrow0 + i0 + ti) * lddq + col0 + d0 + d] = accq[d];
            }
        }
    }
    if (aj < Nk) {
#pragma unroll
        for (int d = 0; d < HH; ++d) {
            dK[(krow0 + aj) * lddkv + col0 + ah * HH + d] = dk[d];
            dV[(krow0 + aj) * lddkv + col0 + ah * HH + d] = dv[d];
        }
    }
}

// ---- the same on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 products): a wave owns a block of 32 query rows and walks the
// key blocks.  Scores are formed TRANSPOSED (rows = keys in the accumulator registers, lane = query), so that the softmax statistics of a
// query are a reduction over registers (+ one half-wave exchange) and dQ += dS^T-block^T . K contracts over accumulator ROWS, which is
// what an accumulator register fed back as the A operand does (register r of lane l is element (row(r, l), l % 32): its two halves are
// the two k-steps of one MFMA).  dV += P^T dO and dK += dS^T Q contract over the other index: P and dS go through a 4-KB wave-private
// LDS transpose first.  dK / dV blocks are accumulated in LDS; wave w visits key block (w + step) % S, with a workgroup barrier per
// step, so no two waves touch a block at the same time and the order of additions is fixed (deterministic).
// Pass 1 computes the statistics online (nothing kept), pass 2 recomputes each score block: 96 MFMAs per (query block, key block) at
// hd = 32 — 10 us per (frame, head) at N = 117 where the VALU kernel above takes 75.
template <int HD>
__global__ __launch_bounds__(512) void k_attn_bwd_mfma(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, int ldkv, const float* __restrict__ O, int ldo,
                                                       const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq,
                                                       float* __restrict__ dK, float* __restrict__ dV, int lddkv, int Nq, int Nk,
                                                       float scale, int QB, int KB) {
    extern __shared__ __attribute__((aligned(16))) float smem_am[];
    constexpr int LD = HD + 1;
    constexpr int DB = (HD + 31) / 32;
    const int NW = blockDim.x >> 6;
    float* Ks = smem_am;
    float* Vs = Ks + (size_t)KB * 32 * LD;
    float* Qs = Vs + (size_t)KB * 32 * LD;
    float* Gs = Qs + (size_t)QB * 32 * LD;            // dO
    float* aK = Gs + (size_t)QB * 32 * LD;            // [KB*32][HD] accumulators
    float* aV = aK + (size_t)KB * 32 * HD;
    float* Ts = aV + (size_t)KB * 32 * HD;            // [NW][32][33]
    float* Ds = Ts + (size_t)NW * 32 * 33;            // [QB*32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hf = lane >> 5;
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t qrow0 = (size_t)b * Nq, krow0 = (size_t)b * Nk;
    const int col0 = h * HD;

    // staging: FOUR iterations' loads are requested before the first LDS store (a rolled load -> store loop is one memory round trip per
    // iteration: 16 + 16 of them were two thirds of this kernel's 47 us at 117 kernels)
    for (int base = 0; base < KB * 32 * HD; base += 4 * blockDim.x) {
        float kv[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * blockDim.x + tid;
            const int j = idx / HD, d = idx - j * HD;
            const bool ok = idx < KB * 32 * HD && j < Nk;
            kv[u] = ok ? Kp[(krow0 + j) * ldkv + col0 + d] : 0.f;
            vv[u] = ok ? Vp[(krow0 + j) * ldkv + col0 + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * blockDim.x + tid;
            if (idx < KB * 32 * HD) {
                const int j = idx / HD, d = idx - j * HD;
                Ks[j * LD + d] = kv[u];
                Vs[j * LD + d] = vv[u];
                aK[idx] = 0.f;
                aV[idx] = 0.f;
            }
        }
    }
    // (the HD lanes of a row sit in one wave, and a wave's last iteration runs whole rows or nothing: QB * 32 * HD and blockDim.x are
    // multiples of 64, so a wave is inside or outside the range as a whole and the shuffles below see full rows)
    for (int base = 0; base < QB * 32 * HD; base += 4 * blockDim.x) {
        float qv[4], gv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * blockDim.x + tid;
            const int i = idx / HD, d = idx - i * HD;
            const bool ok = idx < QB * 32 * HD && i < Nq;
            gv[u] = ok ? dO[(qrow0 + i) * lddo + col0 + d] : 0.f;
            qv[u] = ok ? Q[(qrow0 + i) * ldq + col0 + d] : 0.f;
            ov[u] = ok ? O[(qrow0 + i) * ldo + col0 + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * blockDim.x + tid;
            if (idx < QB * 32 * HD) {
                const int i = idx / HD, d = idx - i * HD;
                Qs[i * LD + d] = qv[u];
                Gs[i * LD + d] = gv[u];
                float pd = gv[u] * ov[u];                                   // D_i = dO_i . O_i: summed over the row's HD lanes
#pragma unroll
                for (int o = 1; o < HD; o <<= 1) pd += __shfl_xor(pd, o);
                if (d == 0) Ds[i] = pd;
            }
        }
    }
    __syncthreads();

    const int i0 = 32 * w;
    const bool mine = w < QB;
    float* T = Ts + (size_t)w * 32 * 33;
    // ---- pass 1: per query (lane column) max and sum of exp over all keys, online over the key blocks; the two half-waves hold
    // different key rows of the same query and are merged at the end
    float mx = -INFINITY, sm = 0.f;
    if (mine) {
        for (int jb = 0; jb < KB; ++jb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < HD / 2; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(32 * jb + li) * LD + 2 * t + hf], Qs[(i0 + li) * LD + 2 * t + hf], acc, 0, 0, 0);
            float bm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * jb + vkn_cd_row(r, lane);
                acc[r] = j < Nk ? acc[r] * scale : -INFINITY;
                bm = fmaxf(bm, acc[r]);
            }
            const float mn = fmaxf(mx, bm);
            if (mn > -INFINITY) {
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) add += __expf(acc[r] - mn);
                sm = sm * __expf(mx - mn) + add;
                mx = mn;
            }
        }
        const float mo = __shfl_xor(mx, 32), so = __shfl_xor(sm, 32);
        const float mn = fmaxf(mx, mo);
        sm = (mx > -INFINITY ? sm * __expf(mx - mn) : 0.f) + (mo > -INFINITY ? so * __expf(mo - mn) : 0.f);
        mx = mn;
    }
    const float inv = (mine && sm > 0.f) ? 1.0f / sm : 0.f;
    const float Di = mine ? Ds[i0 + li] : 0.f;
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

    // ---- pass 2
    const int S = NW;
    for (int step = 0; step < S; ++step) {
        int jb = w + step;
        if (jb >= S) jb -= S;
        if (mine && jb < KB) {
            f32x16 pt, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[r] = dpt[r] = 0.f;
#pragma unroll
            for (int t = 0; t < HD / 2; ++t) {
                const float qv = Qs[(i0 + li) * LD + 2 * t + hf], gv = Gs[(i0 + li) * LD + 2 * t + hf];
                pt = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(32 * jb + li) * LD + 2 * t + hf], qv, pt, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(32 * jb + li) * LD + 2 * t + hf], gv, dpt, 0, 0, 0);
            }
            f32x16 dst;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * jb + vkn_cd_row(r, lane);
                pt[r] = j < Nk ? __expf(pt[r] * scale - mx) * inv : 0.f;
                dst[r] = pt[r] * (dpt[r] - Di) * scale;
            }
            // dQ_i += dS-block (as stored: rows = keys) fed back as A, K rows as B
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = 32 * jb + vkn_cd_row(r, lane);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const float kv = (32 * db + li < HD) ? Ks[jr * LD + 32 * db + li] : 0.f;
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(dst[r], kv, dq[db], 0, 0, 0);
                }
            }
            // P^T-block -> P-block (rows = queries) through the wave's scratch, then dV_j += P^T dO
            f32x16 ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[vkn_cd_row(r, lane) * 33 + li] = pt[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) ps[r] = T[li * 33 + vkn_cd_row(r, lane)];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bool okd = 32 * db + li < HD;
                f32x16 av;
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = okd ? aV[(32 * jb + vkn_cd_row(r, lane)) * HD + 32 * db + li] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float gv = okd ? Gs[(i0 + vkn_cd_row(r, lane)) * LD + 32 * db + li] : 0.f;
                    av = __builtin_amdgcn_mfma_f32_32x32x2f32(ps[r], gv, av, 0, 0, 0);
                }
                if (okd) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) aV[(32 * jb + vkn_cd_row(r, lane)) * HD + 32 * db + li] = av[r];
                }
            }
            // the same for dS: dK_j += dS^T Q
#pragma unroll
            for (int r = 0; r < 16; ++r) T[vkn_cd_row(r, lane) * 33 + li] = dst[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) ps[r] = T[li * 33 + vkn_cd_row(r, lane)];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bool okd = 32 * db + li < HD;
                f32x16 av;
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = okd ? aK[(32 * jb + vkn_cd_row(r, lane)) * HD + 32 * db + li] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float qv = okd ? Qs[(i0 + vkn_cd_row(r, lane)) * LD + 32 * db + li] : 0.f;
                    av = __builtin_amdgcn_mfma_f32_32x32x2f32(ps[r], qv, av, 0, 0, 0);
                }
                if (okd) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) aK[(32 * jb + vkn_cd_row(r, lane)) * HD + 32 * db + li] = av[r];
                }
            }
        }
        __syncthreads();
    }
    if (mine) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + vkn_cd_row(r, lane), d = 32 * db + li;
                if (i < Nq && d < HD) dQ[(qrow0 + i) * lddq + col0 + d] = dq[db][r];
            }
    }
    for (int idx = tid; idx < Nk * HD; idx += blockDim.x) {
        const int j = idx / HD, d = idx - j * HD;
        dK[(krow0 + j) * lddkv + col0 + d] = aK[idx];
        dV[(krow0 + j) * lddkv + col0 + d] = aV[idx];
    }
}

template <int HD>
int attn_bwd_mfma_launch(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo, const float* dO,
                         int lddo, float* dQ, int lddq, float* dK, float* dV, int lddkv, int B, int Nq, int Nk, int heads,
                         hipStream_t st) {
    const int QB = (Nq + 31) / 32, KB = (Nk + 31) / 32;
    const int NW = QB > KB ? QB : KB;
    if (NW > 8) return VKN_E_SHAPE;
    const size_t lds = ((size_t)2 * KB * 32 * (HD + 1) + (size_t)2 * QB * 32 * (HD + 1) + (size_t)2 * KB * 32 * HD + (size_t)NW * 32 * 33 +
                        (size_t)QB * 32) * sizeof(float);
    if (lds > 160 * 1024) return VKN_E_SHAPE;
    if (lds > 64 * 1024) VKN_ALLOW_FULL_LDS(k_attn_bwd_mfma<HD>);
    hipLaunchKernelGGL(k_attn_bwd_mfma<HD>, dim3(heads, B), dim3(64 * NW), lds, st, Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV,
                       lddkv, Nq, Nk, 1.0f / sqrtf((float)HD), QB, KB);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

template <int HD, int TQ>
int attn_bwd_launch_tq(size_t lds, const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo,
                       const float* dO, int lddo, float* dQ, int lddq, float* dK, float* dV, int lddkv, int B, int Nq, int Nk, int heads,
                       hipStream_t st) {
    if (lds > 64 * 1024) VKN_ALLOW_FULL_LDS((k_attn_bwd<HD, TQ>));
    hipLaunchKernelGGL((k_attn_bwd<HD, TQ>), dim3(heads, B), dim3(AB_THREADS), lds, st, Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq,
                       dK, dV, lddkv, Nq, Nk, 1.0f / sqrtf((float)HD));
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

template <int HD>
int attn_bwd_launch(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo, const float* dO,
                    int lddo, float* dQ, int lddq, float* dK, float* dV, int lddkv, int B, int Nq, int Nk, int heads,
                    hipStream_t st) {
    auto lds_of = [&](int tq) {
        return ((size_t)2 * Nk * (HD + 1) + (size_t)2 * tq * (HD + 1) + (size_t)tq * (Nk + 1) + tq) * sizeof(float);
    };
    const size_t cap = 160 * 1024;
    if (lds_of(64) <= cap)
        return attn_bwd_launch_tq<HD, 64>(lds_of(64), Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
    if (lds_of(32) <= cap)
        return attn_bwd_launch_tq<HD, 32>(lds_of(32), Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
    if (lds_of(16) <= cap)
        return attn_bwd_launch_tq<HD, 16>(lds_of(16), Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
    return VKN_E_SHAPE;
}

}  // namespace

extern "C" {

int vkn_linear_dw_f32(const float* dY, int ldy, const float* A, int lda, float* dW, float* db, int M, int K, int Nout, int accumulate,
                      void* stream) {
    if (!dY || !A || !dW || M <= 0 || K <= 0 || Nout <= 0 || ldy < Nout || lda < K) return VKN_E_ARG;
    VKN_ALLOW_FULL_LDS(k_gemm_tn);
    hipLaunchKernelGGL(k_gemm_tn, dim3((Nout + 31) / 32, (K + 31) / 32), dim3(TN_THREADS), TN_LDS, static_cast<hipStream_t>(stream), dY,
                       ldy, A, lda, dW, K, db, M, Nout, K, accumulate);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_split_weights_batch_f32(const VknSplitItem* items, int nitems, void* stream) {
    if (!items || nitems <= 0 || nitems > VKN_SPLIT_MAX_ITEMS) return VKN_E_ARG;
    SplitTab T;
    int tiles = 0;
    for (int i = 0; i < nitems; ++i) {
        const VknSplitItem& it = items[i];
        if (!it.W || !it.images || it.Nout <= 0 || it.K <= 0 || it.kvalid < 0 || it.kvalid > it.K) return VKN_E_ARG;
        if (it.K % 32 != 0 || (reinterpret_cast<uintptr_t>(it.images) & 15) != 0) return VKN_E_SHAPE;
        T.W[i] = it.W;
        T.dst[i] = static_cast<__bf16*>(it.images);
        T.ldn[i] = it.ldn;
        T.ldk[i] = it.ldk;
        T.nout[i] = it.Nout;
        T.k[i] = it.K;
        T.kvalid[i] = it.kvalid;
        T.tile0[i] = tiles;
        tiles += ((it.Nout + 255) / 256) * (it.K / 32);
    }
    T.tile0[nitems] = tiles;
    hipLaunchKernelGGL(k_split_batch, dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), T, nitems);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

size_t vkn_sizeof_split_item(void) { return sizeof(VknSplitItem); }

size_t vkn_sizeof_dw_item(void) { return sizeof(VknDwItem); }
size_t vkn_sizeof_updator_norms(void) { return sizeof(VknUpdatorNorms); }
size_t vkn_sizeof_updator_norm_grads(void) { return sizeof(VknUpdatorNormGrads); }

int vkn_linear_dw_batch_f32(const VknDwItem* items, int nitems, int M, void* stream) {
    if (!items || nitems <= 0 || nitems > VKN_DW_MAX_ITEMS || M <= 0) return VKN_E_ARG;
    TnTab T;
    int tiles = 0;
    for (int i = 0; i < nitems; ++i) {
        const VknDwItem& it = items[i];
        if (!it.dY || !it.A || !it.dW || it.Nout <= 0 || it.K <= 0 || it.ldy < it.Nout || it.lda < it.K) return VKN_E_ARG;
        T.Y[i] = it.dY; T.A[i] = it.A; T.dW[i] = it.dW; T.db[i] = it.db;
        T.ldy[i] = it.ldy; T.lda[i] = it.lda; T.nout[i] = it.Nout; T.k[i] = it.K;
        T.tile0[i] = tiles;
        tiles += ((it.Nout + 63) / 64) * ((it.K + 127) / 128);
    }
    T.tile0[nitems] = tiles;
    VKN_ALLOW_FULL_LDS(k_gemm_tn_batch);
    hipLaunchKernelGGL(k_gemm_tn_batch, dim3(tiles), dim3(TL_THREADS), TL_LDS, static_cast<hipStream_t>(stream), T, nitems, M, tiles);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_layernorm_act_fwd_f32(const float* in, int ldi, const float* resid, int ldr, const float* gamma, const float* beta, float eps,
                              int act, float* out, int ldo, float* stats, int M, int C, void* stream) {
    if (!in || !out || M <= 0 || C <= 0 || ldi < C || ldo < C || (resid && ldr < C) || act < 0 || act > 2) return VKN_E_ARG;
    if (C > 64 * LN_MAXV) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_ln_fwd, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, resid, ldr, gamma, beta,
                       eps, act, out, ldo, stats, M, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_layernorm_act_bwd_f32(const float* dy, int lddy, const float* in, int ldi, const float* resid, int ldr, const float* gamma,
                              const float* beta, const float* stats, int act, float* dx, int lddx, float* dgamma, float* dbeta, int M,
                              int C, void* stream) {
    if (!dy || !in || !stats || !dx || M <= 0 || C <= 0 || lddy < C || ldi < C || lddx < C || (resid && ldr < C) || act < 0 || act > 2)
        return VKN_E_ARG;
    if (C > 64 * LN_MAXV) return VKN_E_SHAPE;
    const int nrb = (M + LNB_ROWS - 1) / LNB_ROWS;
    const int ncb = (dgamma || dbeta) ? (C + 31) / 32 : 0;
    hipLaunchKernelGGL(k_ln_bwd, dim3(nrb + ncb), dim3(LNB_THREADS), 0, static_cast<hipStream_t>(stream), dy, lddy, in, ldi, resid, ldr,
                       gamma, beta, stats, act, dx, lddx, dgamma, dbeta, M, C, nrb);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_updator_gate_product_f32(const float* params, const float* inputs, float* gate_feats, int M, int C, void* stream) {
    if (!params || !inputs || !gate_feats || M <= 0 || C <= 0) return VKN_E_ARG;
    if (C % 4) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_gprod_fwd, dim3((M * (C / 4) + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), params, inputs, 2 * C,
                       gate_feats, M, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_updator_gate_product_bwd_f32(const float* d_gate_feats, const float* params, const float* inputs, float* d_params, float* d_inputs,
                                     int M, int C, void* stream) {
    if (!d_gate_feats || !params || !inputs || !d_params || !d_inputs || M <= 0 || C <= 0) return VKN_E_ARG;
    if (C % 4) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_gprod_bwd, dim3((M * (C / 4) + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), d_gate_feats, params,
                       inputs, 2 * C, d_params, d_inputs, M, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

static bool upd_norms_ok(const VknUpdatorNorms* n) {
    return n && n->norm_in_w && n->norm_in_b && n->norm_out_w && n->norm_out_b && n->input_norm_in_w && n->input_norm_in_b &&
           n->input_norm_out_w && n->input_norm_out_b;
}

int vkn_updator_mix_fwd_f32(const float* gates, const float* params, const float* inputs, const VknUpdatorNorms* norms, float eps,
                            float* features, float* stats, int M, int C, void* stream) {
    if (!gates || !params || !inputs || !features || !stats || !upd_norms_ok(norms) || M <= 0 || C <= 0) return VKN_E_ARG;
    if (C > 64 * LN_MAXV) return VKN_E_SHAPE;
    const UpdNorms nw{norms->norm_in_w, norms->norm_in_b, norms->norm_out_w, norms->norm_out_b, norms->input_norm_in_w,
                      norms->input_norm_in_b, norms->input_norm_out_w, norms->input_norm_out_b, norms->input_gate_b, norms->update_gate_b};
    hipLaunchKernelGGL(k_mix_fwd, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), gates, params, inputs, 2 * C, nw, eps,
                       features, stats, M, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_updator_mix_bwd_f32(const float* d_features, const float* gates, const float* params, const float* inputs,
                            const VknUpdatorNorms* norms, const float* stats, float* d_gates, float* d_params, float* d_inputs,
                            const VknUpdatorNormGrads* d_norms, int M, int C, void* stream) {
    if (!d_features || !gates || !params || !inputs || !stats || !d_gates || !d_params || !d_inputs || !upd_norms_ok(norms) || M <= 0 ||
        C <= 0)
        return VKN_E_ARG;
    if (C > 64 * LN_MAXV) return VKN_E_SHAPE;
    const UpdNorms nw{norms->norm_in_w, norms->norm_in_b, norms->norm_out_w, norms->norm_out_b, norms->input_norm_in_w,
                      norms->input_norm_in_b, norms->input_norm_out_w, norms->input_norm_out_b, norms->input_gate_b, norms->update_gate_b};
    UpdNormGrads gw{};
    if (d_norms)
        gw = UpdNormGrads{d_norms->norm_in_w, d_norms->norm_in_b, d_norms->norm_out_w, d_norms->norm_out_b, d_norms->input_norm_in_w,
                          d_norms->input_norm_in_b, d_norms->input_norm_out_w, d_norms->input_norm_out_b};
    const int nrb = (M + LNB_ROWS - 1) / LNB_ROWS;
    const int ncb = d_norms ? (C + 31) / 32 : 0;
    hipLaunchKernelGGL(k_mix_bwd, dim3(nrb + ncb), dim3(LNB_THREADS), 0, static_cast<hipStream_t>(stream), d_features, gates, params,
                       inputs, 2 * C, nw, stats, d_gates, d_params, d_inputs, gw, M, C, nrb);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_attention_f32(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* out, int ldo, int B, int Nq, int Nk,
                      int heads, int hd, void* stream) {
    if (!Q || !K || !V || !out || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0) return VKN_E_ARG;
    return vkn_launch_attn(Q, ldq, K, V, ldkv, out, ldo, B, Nq, Nk, heads, hd, static_cast<hipStream_t>(stream));
}

int vkn_attention_bwd_f32(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo, const float* dO,
                          int lddo, float* dQ, int lddq, float* dK, float* dV, int lddkv, int B, int Nq, int Nk, int heads, int hd,
                          void* stream) {
    if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0) return VKN_E_ARG;
    if (Nk > AB_THREADS / 2) return VKN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {   // the matrix-core kernel where its LDS footprint fits (every shipped shape up to N = 128 at hd = 32, N = 224 at hd = 16)
        int rc = VKN_E_SHAPE;
        if (hd == 16) rc = attn_bwd_mfma_launch<16>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        else if (hd == 32) rc = attn_bwd_mfma_launch<32>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        else if (hd == 64) rc = attn_bwd_mfma_launch<64>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        if (rc != VKN_E_SHAPE) return rc;
    }
    switch (hd) {
        case 4: return attn_bwd_launch<4>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        case 8: return attn_bwd_launch<8>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        case 16: return attn_bwd_launch<16>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        case 32: return attn_bwd_launch<32>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        case 64: return attn_bwd_launch<64>(Q, ldq, K, V, ldkv, O, ldo, dO, lddo, dQ, lddq, dK, dV, lddkv, B, Nq, Nk, heads, st);
        default: return VKN_E_SHAPE;
    }
}

}  // extern "C"
