// vkn_update.hip — the [B*N, C] "kernel update + interaction" block of one stage, plus the final bilinear upsample.
//
// Covers (reference file:line):
//   KernelUpdator.forward                      knet/kernel_updator.py:56-93
//   attention + attention_norm                 knet/det/kernel_update_head.py:204-208   (mmcv MultiheadAttention)
//   ffn + ffn_norm                             knet/det/kernel_update_head.py:214-215   (mmcv FFN)
//   cls_fcs / fc_cls / mask_fcs / fc_mask      knet/det/kernel_update_head.py:217-227
//   video tracking link (previous_type="ffn")  knet/video/kernel_update_head.py:394-415
//   F.interpolate(x mask_upsample_stride)      knet/det/kernel_iter_head.py:122-130
//
// Building blocks:
//   k_gemm      out = epi(A . W^T): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain), 32-row x 256-col tile per
//               512-thread workgroup (wave w = column block w), A/W tiles staged in padded LDS with register prefetch of the
//               next K-tile, optional split-K over blockIdx.z; optional elementwise-product A prologue (the updator's gate).
//               The accumulator tile goes through LDS to a row-wise epilogue (one wave per row).
//   row epilogue  + bias (optionally scaled per row: the folded feat_transform bias x pixel count) + residual -> LayerNorm
//               -> ReLU/sigmoid -> store; optional dot-product side output (folded decode bias) and f16 hi/lo plane output
//               (the decode kernels' LDS image).  Shared by k_gemm (fused) and k_rowepi (after split-K).
//   k_ku_mix    f = ug * LN(param_out) + ig * LN(input_out)              (knet/kernel_updator.py:79-88)
//   k_attn      softmax(q k^T / sqrt(d)) v per (frame, head), one wave per query row.
//   k_upsample  bilinear, align_corners=False, integer scale.
#include "vkn_common.h"
#include "vkn_launch.h"


#define GM_THREADS 512
#define GM_BM 32
#define GM_BN 256
#define GM_KT 32
#define GM_LDA 33
#define GM_LDT 260

// Per-lane column constants of the row epilogue (lane owns columns lane + 64*q): loaded ONCE per wave with clamped
// addresses and no per-lane branches, so the loads issue back-to-back instead of one L2 round trip each.
struct VknEpiCols {
    float bias[4], bias2[4], lnw[4], lnb[4], dot[4];
    int cidx[4];  // clamped absolute column
    bool ok[4];
    bool do_ln;
};

__device__ __forceinline__ void vkn_epi_load_cols(const VknEpi& e, int ncols, int col0, int lane, VknEpiCols& c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cl = lane + 64 * q;
        c.ok[q] = cl < ncols;
        c.cidx[q] = col0 + min(cl, ncols - 1);
        c.bias[q] = 0.f; c.bias2[q] = 0.f; c.lnw[q] = 1.f; c.lnb[q] = 0.f; c.dot[q] = 0.f;
    }
    if (e.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c.bias[q] = e.bias[c.cidx[q]];
    }
    if (e.bias2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c.bias2[q] = e.bias2[c.cidx[q]];
    }
    c.do_ln = (e.ln_w != nullptr) && (col0 >= e.ln_from_col);
    if (c.do_ln) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c.lnw[q] = e.ln_w[c.cidx[q] - e.ln_from_col];
            c.lnb[q] = e.ln_b[c.cidx[q] - e.ln_from_col];
        }
    }
    if (e.dot_vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c.dot[q] = e.dot_vec[c.cidx[q]];
    }
}

// Per-row global operands of the epilogue (residual row, bias scale).  Loaded for ALL rows of a wave before the first row is
// processed: inside the row loop they sit behind the previous row's stores (which may alias them as far as the compiler knows),
// i.e. one L2 round trip per row on the critical path of every GEMM launch.
struct VknEpiRow {
    float rv[4];
    float bs;
};
__device__ __forceinline__ void vkn_epi_load_row(const VknEpi& e, const VknEpiCols& c, int row, VknEpiRow& r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) r.rv[q] = 0.f;
    if (e.resid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r.rv[q] = e.resid[(size_t)row * e.ldr + c.cidx[q]];
    }
    r.bs = (e.bias && e.rowscale) ? e.rowscale[row] : 1.f;
}

// Row-wise epilogue executed by ONE wave for ONE row.  v[q] = accumulated value at column lane + 64*q.
__device__ __forceinline__ void vkn_row_epilogue(const VknEpi& e, const VknEpiCols& c, int row, int ncols, int lane,
                                                 float (&v)[4], const VknEpiRow& pre) {
    const float (&rv)[4] = pre.rv;
    const float bs = pre.bs;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = c.ok[q] ? (v[q] + c.bias[q] * bs + c.bias2[q] + rv[q]) : 0.f;
    if (c.do_ln) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += v[q];
        const float mean = vkn_wave_sum(s) / (float)ncols;
        float d2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = c.ok[q] ? (v[q] - mean) : 0.f;
            d2 += d * d;
        }
        const float rstd = 1.0f / sqrtf(vkn_wave_sum(d2) / (float)ncols + e.eps);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (v[q] - mean) * rstd * c.lnw[q] + c.lnb[q];
    }
    if (e.act == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
    } else if (e.act == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = 1.0f / (1.0f + expf(-v[q]));
    }
    if (e.out) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c.ok[q]) e.out[(size_t)row * e.ldo + c.cidx[q]] = v[q];
    }
    if (e.dot_vec) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += c.ok[q] ? v[q] * c.dot[q] : 0.f;
        s = vkn_wave_sum(s);
        if (lane == 0) e.dot_out[row] = s + (e.dot_bias ? *e.dot_bias : 0.f);
    }
    if (e.plane_hi) {
        const int b = row / e.rows_per_frame, n = row - b * e.rows_per_frame;
        const size_t base = ((size_t)b * e.NPT + n) * e.ldo;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c.ok[q]) {
                _Float16 h, l;
                vkn_split_f16(v[q], h, l);
                e.plane_hi[base + c.cidx[q]] = h;
                e.plane_lo[base + c.cidx[q]] = l;
            }
    }
}

// out[M][Nout] (tile 32 x 256) = A[M][K] . W[Nout][K]^T ; A2 != null: A := A (.) A2 elementwise.
// gridDim = (ceil(Nout/256), ceil(M/32), ksplit).  ksplit > 1: raw partial sums to `partial` [ks][M][Nout].
__global__ __launch_bounds__(GM_THREADS, 2) void k_gemm(const float* __restrict__ A, const float* __restrict__ A2,
                                                        const float* __restrict__ A3, const float* __restrict__ A4, int lda,
                                                        const float* __restrict__ W, int M, int K, int Nout,
                                                        float* __restrict__ partial, VknEpi epi) {
    __shared__ __attribute__((aligned(16))) float smem[GM_BM * GM_LDA + GM_BN * GM_LDA];
    float* As = smem;                   // [32][33]
    float* Ws = smem + GM_BM * GM_LDA;  // [256][33]; reused as the [32][260] output tile (8320 <= 8448 floats)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int g = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    const int ksplit = gridDim.z;
    const int ktiles = K / GM_KT;
    const int kt_per = (ktiles + ksplit - 1) / ksplit;
    const int kt_begin = blockIdx.z * kt_per;
    const int kt_end = min(ktiles, kt_begin + kt_per);

    // staging roles: thread -> (row = tid>>5 [+16*i], k = tid&31)
    const int sk = tid & 31, sr = tid >> 5;  // sr in 0..15
    // Prefetch loads are unconditional on clamped rows (no per-lane branches) and nothing is consumed before the
    // stash, so a whole K-tile (20 loads per thread) stays in flight behind the MFMAs of the previous tile.
    const float* A2p = A2 ? A2 : A;
    const float* A3p = A3 ? A3 : A;
    const float* A4p = A4 ? A4 : A;
    const bool mul = (A2 != nullptr), two = (A3 != nullptr);
    float ra0, ra1, rb0, rb1, rc0, rc1, rd0, rd1;
    float rw[16];
    const size_t aoff0 = (size_t)min(m0 + sr, M - 1) * lda + sk;
    const size_t aoff1 = (size_t)min(m0 + sr + 16, M - 1) * lda + sk;
    const bool aok0 = (m0 + sr) < M, aok1 = (m0 + sr + 16) < M;

#define GM_FETCH(KT)                                                                             \
    do {                                                                                         \
        const int kof_ = (KT) * GM_KT;                                                           \
        ra0 = A[aoff0 + kof_];                                                                   \
        ra1 = A[aoff1 + kof_];                                                                   \
        rb0 = A2p[aoff0 + kof_];                                                                 \
        rb1 = A2p[aoff1 + kof_];                                                                 \
        rc0 = A3p[aoff0 + kof_];                                                                 \
        rc1 = A3p[aoff1 + kof_];                                                                 \
        rd0 = A4p[aoff0 + kof_];                                                                 \
        rd1 = A4p[aoff1 + kof_];                                                                 \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                           \
            rw[i] = W[(size_t)min(n0 + sr + 16 * i, Nout - 1) * K + kof_ + sk];                  \
    } while (0)

#define GM_STASH()                                                                               \
    do {                                                                                         \
        As[sr * GM_LDA + sk] = aok0 ? ((mul ? ra0 * rb0 : ra0) + (two ? rc0 * rd0 : 0.f)) : 0.f; \
        As[(sr + 16) * GM_LDA + sk] = aok1 ? ((mul ? ra1 * rb1 : ra1) + (two ? rc1 * rd1 : 0.f)) : 0.f; \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                           \
            Ws[(sr + 16 * i) * GM_LDA + sk] = (n0 + sr + 16 * i < Nout) ? rw[i] : 0.f;           \
    } while (0)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool active = (n0 + wave * 32) < Nout;  // this wave's column block has at least one real column (uniform)

    if (kt_begin < kt_end) GM_FETCH(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        GM_STASH();
        __syncthreads();
        if (kt + 1 < kt_end) GM_FETCH(kt + 1);
        if (active) {
            const float* ap = As + li * GM_LDA + g;
            const float* bp = Ws + (wave * 32 + li) * GM_LDA + g;
#pragma unroll
            for (int kk = 0; kk < GM_KT / 2; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bp[2 * kk], acc, 0, 0, 0);
        }
        __syncthreads();
    }
#undef GM_FETCH
#undef GM_STASH

    if (ksplit > 1) {
        // raw partial tile -> global [ks][M][Nout]; lanes 0..31 write 128 contiguous bytes
        float* pz = partial + (size_t)blockIdx.z * M * Nout;
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + vkn_cd_row(r, lane), col = n0 + wave * 32 + li;
                if (row < M && col < Nout) pz[(size_t)row * Nout + col] = acc[r];
            }
        }
        return;
    }

    // ---- fused epilogue: accumulators -> LDS tile [32][260] -> one wave per row
    float* T = Ws;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[vkn_cd_row(r, lane) * GM_LDT + wave * 32 + li] = active ? acc[r] : 0.f;
    __syncthreads();
    const int ncols = min(GM_BN, Nout - n0);
    VknEpiCols cols;
    vkn_epi_load_cols(epi, ncols, n0, lane, cols);
    VknEpiRow pre[GM_BM / 8];
#pragma unroll
    for (int i = 0; i < GM_BM / 8; ++i) vkn_epi_load_row(epi, cols, min(m0 + wave * (GM_BM / 8) + i, M - 1), pre[i]);
#pragma unroll
    for (int i = 0; i < GM_BM / 8; ++i) {
        const int rl = wave * (GM_BM / 8) + i, row = m0 + rl;
        if (row < M) {  // uniform
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = T[rl * GM_LDT + lane + 64 * q];
            vkn_row_epilogue(epi, cols, row, ncols, lane, v, pre[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ bf16x3 split GEMM
// Same tile geometry, epilogue and split-K protocol as k_gemm, but the contraction runs on v_mfma_f32_32x32x16_bf16 with
// BOTH operands split into three bf16 terms (x = h + m + l exactly to 2^-24: full fp32 range, unlike f16) and the six
// significant cross products accumulated in fp32:  h*h + h*m + m*h + m*m + h*l + l*h   (dropped terms <= 2^-24 relative).
// 6 MFMAs per 16 k at the bf16 rate = 2.7x the fp32-MFMA rate at fp32-class accuracy.
//
// The kernel-update chain is a sequence of short dependent kernels on M ~ 10^3 rows, so what matters is the serial latency of
// one launch.  tools/gemm_trace.sh showed the per-K-tile phases {L1 fill of the 48 KB weight tile, LDS write, LDS read + MFMA,
// two barriers} each cost 0.3-0.5 us and serialised (1.2 us per K-tile).  This version overlaps them:
//   * weights are PRE-SPLIT (k_split_w3, once per weight update) into the exact LDS image of a (256-column, 32-k) tile —
//     Wp[col tile][k tile][plane 3][q 4][row 256][8 bf16], 48 KB contiguous — and stream global -> LDS with
//     `global_load_lds_dwordx4` (no VGPR round trip, no ds_write); a fragment read (fixed plane/q, 32 consecutive rows) is
//     32 consecutive 16-B slots: conflict-free;
//   * A [M][lda] fp32 is split in registers while it is staged (4 KB per tile);
//   * LDS is double-buffered: the DMA + A loads of tile t+1 are issued before the MFMAs of tile t; ONE barrier per tile.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define GS_LDR 40                   // bf16 per LDS row of the A image: 32 + 8 pad (80 B: conflict-free ds_read_b128)
#define GS_WTILE (3 * 4 * 256 * 8)  // bf16 elements of one weight tile image (48 KB)
#define GS_ATILE (3 * GM_BM * GS_LDR)

__device__ __forceinline__ void vkn_split_bf16x3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// fp32 W [Nout][K] -> tile images Wp[ceil(Nout/256)][K/32][3][4][256][8] (rows >= Nout zero).  grid = (K/32, ceil(Nout/256)).
// Element (n, k) of the weight is read at W[n * ldn + k * ldk]: (K, 1) for a torch Linear weight, (1, Nout) for the images of its
// TRANSPOSE (the dA = dY . W GEMM of the backward pass: "weight" [K_fwd][Nout_fwd] read from the same storage).
__global__ __launch_bounds__(256) void k_split_w3(const float* __restrict__ W, __bf16* __restrict__ Wp, int Nout, int K, size_t ldn,
                                                  size_t ldk) {
    const int kt = blockIdx.x, nt = blockIdx.y;
    __bf16* dst = Wp + ((size_t)nt * gridDim.x + kt) * GS_WTILE;
    const int row = threadIdx.x, n = nt * 256 + row;
    for (int k = 0; k < 32; ++k) {
        __bf16 h = (__bf16)0.f, m = (__bf16)0.f, l = (__bf16)0.f;
        if (n < Nout) vkn_split_bf16x3(W[(size_t)n * ldn + (size_t)(kt * 32 + k) * ldk], h, m, l);
        const int q = k >> 3, e = k & 7;
        dst[((0 * 4 + q) * 256 + row) * 8 + e] = h;
        dst[((1 * 4 + q) * 256 + row) * 8 + e] = m;
        dst[((2 * 4 + q) * 256 + row) * 8 + e] = l;
    }
}

// Up to two independent problems (same M, K) run as ONE launch, selected by blockIdx.z when ksplit == 1: pairing independent
// layers (dynamic/input layer, the two gates, the cls/mask branches) removes whole launch + prologue + epilogue latencies
// from the critical path of the chain.
// ABL (debug build, VKN_GEMM_ABL; WRONG results by construction, time attribution only): 1 = no K loop, 2 = no row epilogue
template <int ABL>
__global__ __launch_bounds__(GM_THREADS) void k_gemm_s3(VknGemmProb p0, VknGemmProb p1, int nprob, int M, int K,
                                                           float* __restrict__ partial) {
    const bool second = (nprob > 1) && (blockIdx.z == 1);
    const float* __restrict__ A = second ? p1.A : p0.A;
    const float* __restrict__ A2 = second ? p1.A2 : p0.A2;
    const float* __restrict__ A3 = second ? p1.A3 : p0.A3;
    const float* __restrict__ A4 = second ? p1.A4 : p0.A4;
    const int lda = second ? p1.lda : p0.lda;
    const __bf16* __restrict__ Wp = static_cast<const __bf16*>(second ? p1.Wsplit : p0.Wsplit);
    const int Nout = second ? p1.Nout : p0.Nout;
    const VknEpi epi = second ? p1.epi : p0.epi;
    extern __shared__ __attribute__((aligned(16))) char smem_s3[];
    __bf16* Wl = reinterpret_cast<__bf16*>(smem_s3);  // [3][GS_WTILE]; buffer 0 is reused as the fp32 [32][260] output tile
    __bf16* Al = Wl + 3 * GS_WTILE;                   // [2][3][32][40]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    if (n0 >= Nout) return;  // grouped launch: the grid is sized for the wider problem (uniform exit before any barrier)
    const int ksplit = (nprob > 1) ? 1 : gridDim.z;
    const int kz = (nprob > 1) ? 0 : blockIdx.z;
    const int ktiles = K >> 5;
    const int kt_per = (ktiles + ksplit - 1) / ksplit;
    const int kt_begin = kz * kt_per;
    const int kt_end = min(ktiles, kt_begin + kt_per);

    // Roles (wave-uniform): waves 0-3 issue the weight DMA, waves 4-7 fetch / split / stash the A tile.  The split matters for the
    // waits: hipcc waits vmcnt(0) for ANY register load once LDS-DMA instructions are pending in the same wave (measured: it did
    // so with the A loads strictly older than the DMA), which would drain the prefetch every K-tile.  With the roles apart, the DMA
    // waves carry no compiler-visible dependency and are held only by the explicit vmcnt below; the A waves have nothing else in flight.
    const bool a_role = wave >= 4;
    const int at = tid & 255, ar = (at >> 3) & 31, aq = at & 7;
    const size_t aoff = (size_t)min(m0 + ar, M - 1) * lda + 4 * aq;
    const float* A2p = A2 ? A2 : A;
    const float* A3p = A3 ? A3 : A;
    const float* A4p = A4 ? A4 : A;
    const bool mul = (A2 != nullptr), two = (A3 != nullptr);
    const __bf16* wtile0 = Wp + (size_t)blockIdx.x * ktiles * GS_WTILE;  // this column tile's images, K-tile major
    f32x4 S0[4], S1[4];  // two register sets of A fragments (tile i + 1 waiting to be stashed, tile i + 2 in flight)

    // weight tile KT -> LDS buffer BUF: 3072 x 16 B = 12 DMA instructions per thread of waves 0-3; piece index = i*256 + tid, so
    // every wave writes 64 consecutive slots (the DMA's LDS address is wave base + lane*16) from 1 KB of contiguous global memory
#define GS_DMA(KT, BUF)                                                                                               \
    do {                                                                                                              \
        const char* gsrc_ = reinterpret_cast<const char*>(wtile0 + (size_t)(KT) * GS_WTILE) + (size_t)at * 16;        \
        char* ldst_ = reinterpret_cast<char*>(Wl + (size_t)(BUF) * GS_WTILE) + (size_t)wave * 1024;                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_)                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_ + i_ * 4096),      \
                                             (__attribute__((address_space(3))) void*)(ldst_ + i_ * 4096), 16, 0, 0); \
    } while (0)

#define GS_AFETCH(KT, S)                                                            \
    do {                                                                            \
        S[0] = *reinterpret_cast<const f32x4*>(A + aoff + (size_t)(KT) * 32);       \
        S[1] = *reinterpret_cast<const f32x4*>(A2p + aoff + (size_t)(KT) * 32);     \
        S[2] = *reinterpret_cast<const f32x4*>(A3p + aoff + (size_t)(KT) * 32);     \
        S[3] = *reinterpret_cast<const f32x4*>(A4p + aoff + (size_t)(KT) * 32);     \
    } while (0)

#define GS_ASTASH(BUF, S)                                                                         \
    do {                                                                                          \
        bf16x4 h_, m_, l_;                                                                        \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
            const float v_ = (mul ? S[0][e] * S[1][e] : S[0][e]) + (two ? S[2][e] * S[3][e] : 0.f); \
            __bf16 hh_, mm_, ll_;                                                                 \
            vkn_split_bf16x3(v_, hh_, mm_, ll_);                                                  \
            h_[e] = hh_;                                                                          \
            m_[e] = mm_;                                                                          \
            l_[e] = ll_;                                                                          \
        }                                                                                         \
        __bf16* d_ = Al + (size_t)(BUF) * GS_ATILE + ar * GS_LDR + 4 * aq;                        \
        *reinterpret_cast<bf16x4*>(d_) = h_;                                                      \
        *reinterpret_cast<bf16x4*>(d_ + GM_BM * GS_LDR) = m_;                                     \
        *reinterpret_cast<bf16x4*>(d_ + 2 * GM_BM * GS_LDR) = l_;                                 \
    } while (0)

#define GS_MFMA(I)                                                                                                    \
    do {                                                                                                              \
        { /* every wave multiplies (image rows >= Nout are zero): no branch in the loop body */                       \
            const __bf16* Ab = Al + (size_t)((I) & 1) * GS_ATILE;                                                     \
            const __bf16* Wb = Wl + (size_t)((I) % 3) * GS_WTILE;                                                     \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                        \
                const __bf16* ap = Ab + li * GS_LDR + (ks << 4) + (g << 3);                                           \
                const __bf16* bp = Wb + (((ks << 1) + g) * 256 + wave * 32 + li) * 8; /* plane 0, q = 2*ks + g */     \
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);                                               \
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + GM_BM * GS_LDR);                              \
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * GM_BM * GS_LDR);                          \
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp);                                               \
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bp + 4 * 256 * 8);                                 \
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + 8 * 256 * 8);                                 \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0); /* smallest terms first */       \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);                                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);                                  \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool active = (n0 + wave * 32) < Nout;

    // per-lane column constants of the row epilogue (bias / LayerNorm weights / dot vector): requested BEFORE the K loop, so their
    // L2 round trips ride under it instead of sitting between the last MFMA and the first output row
    const int ncols = min(GM_BN, Nout - n0);
    VknEpiCols cols;
    vkn_epi_load_cols(epi, ncols, n0, lane, cols);
    VknEpiRow pre[GM_BM / 8];  // ... and so are the per-row operands (residual rows, bias scales) of this wave's four output rows
#pragma unroll
    for (int i = 0; i < GM_BM / 8; ++i) vkn_epi_load_row(epi, cols, min(m0 + wave * (GM_BM / 8) + i, M - 1), pre[i]);

    // Three weight buffers: while tile i is multiplied, tiles i+1 AND i+2 are in flight (96 KB per CU); the A fragments run two
    // tiles ahead as well (tile i+2 requested before the MFMAs of tile i, tile i+1 split and written after them).  Time attribution
    // (debug build, VKN_GEMM_ABL, B = 1): the K loop cost 8.4 of a launch's 15 us with A one tile ahead — its L2 round trip
    // (~0.7 us) did not fit under the 0.3 us of MFMAs of one tile — and the depth of the WEIGHT ring made no difference by itself.
    // The DMA -> LDS dependency is invisible to the compiler, so the DMA waves' waits are explicit: vmcnt counts in order, the 12
    // DMA instructions of tile i+2 are the youngest, vmcnt(12) = "tile i+1 has landed".  Both roles run the same number of barriers.
    // BAR: this wave's LDS writes are done (lgkmcnt) + workgroup barrier, as ONE asm statement: behind a __syncthreads() hipcc
    // strengthens the wait to vmcnt(0) (its workgroup release fence), which would drain the prefetch.
#define GS_BAR(VM) asm volatile("s_waitcnt " VM "lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int nkt = VKN_ABL_IS(ABL, 1) ? 0 : kt_end - kt_begin;
    if (nkt > 0) {
        const int klast = kt_end - 1;
        if (a_role) {
            GS_AFETCH(kt_begin, S0);
            GS_AFETCH(min(kt_begin + 1, klast), S1);
            GS_ASTASH(0, S0);
            GS_BAR("");
            for (int j = 0; j + 1 < nkt; j += 2) {
                GS_AFETCH(min(kt_begin + j + 2, klast), S0);  // past the end: re-reads the last tile, never stashed into a live buffer
                GS_MFMA(j);
                GS_ASTASH(1, S1);
                GS_BAR("");
                if (j + 2 < nkt) {
                    GS_AFETCH(min(kt_begin + j + 3, klast), S1);
                    GS_MFMA(j + 1);
                    GS_ASTASH(0, S0);
                    GS_BAR("");
                }
            }
        } else {
            GS_DMA(kt_begin, 0);
            if (nkt > 1) {
                GS_DMA(kt_begin + 1, 1);
                GS_BAR("vmcnt(12) ");
            } else {
                GS_BAR("vmcnt(0) ");
            }
            for (int i = 0; i + 1 < nkt; ++i) {
                if (i + 2 < nkt) {
                    GS_DMA(kt_begin + i + 2, (i + 2) % 3);  // that buffer held tile i-1: every wave passed the barrier after reading it
                    GS_MFMA(i);
                    GS_BAR("vmcnt(12) ");  // tile i+1 landed; tile i+2 may still be in flight
                } else {
                    GS_MFMA(i);
                    GS_BAR("vmcnt(0) ");
                }
            }
        }
        GS_MFMA(nkt - 1);
        __syncthreads();  // buffer 0 becomes the output tile below
    }
#undef GS_BAR
#undef GS_DMA
#undef GS_AFETCH
#undef GS_ASTASH
#undef GS_MFMA

    if (ksplit > 1) {
        float* pz = partial + (size_t)kz * M * Nout;
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + vkn_cd_row(r, lane), col = n0 + wave * 32 + li;
                if (row < M && col < Nout) pz[(size_t)row * Nout + col] = acc[r];
            }
        }
        return;
    }

    float* T = reinterpret_cast<float*>(Wl);  // [32][260] fp32 = 33,280 B <= 49,152 B (weight buffer 0; all reads are done)
#pragma unroll
    for (int r = 0; r < 16; ++r) T[vkn_cd_row(r, lane) * GM_LDT + wave * 32 + li] = active ? acc[r] : 0.f;
    __syncthreads();
    if (VKN_ABL_IS(ABL, 2)) {
        if (T[tid] == 12345.678f) partial[0] = cols.bias[0];  // keep the loads alive
        return;
    }
#pragma unroll
    for (int i = 0; i < GM_BM / 8; ++i) {
        const int rl = wave * (GM_BM / 8) + i, row = m0 + rl;
        if (row < M) {  // uniform
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = T[rl * GM_LDT + lane + 64 * q];
            vkn_row_epilogue(epi, cols, row, ncols, lane, v, pre[i]);
        }
    }
}

#ifdef VKN_DEBUG  // rejected / time-attribution variants live outside the product sources
#include "../../tools/experiments/gemm_variants.inc"
#endif

// ------------------------------------------------------------------------------------------------ fused FFN
// partial[hs] = relu(X . W1[hidden range hs]^T + b1) . W2[:, hidden range hs]^T         (mmcv FFN, both Linears in one kernel)
// The [M x 2048] hidden activations never leave the chip: workgroup (row tile, hidden split hs) walks its 256-wide hidden chunks;
// per chunk it runs GEMM 1 (K = C, tile images of W1's column tile = the chunk) exactly like k_gemm_s3, turns the accumulators
// into ReLU(. + b1) split to bf16x3 straight into an LDS image, and uses that image as the A operand of GEMM 2 (K = the chunk's 256
// hidden units = 8 K-tiles of W2's images), accumulating the [32 x C] output over its chunks.  The hidden splits are summed in
// fixed order by k_rowepi (+ b2, residual, LayerNorm).  Replaces two launches, the M x 2048 round trip through HBM and half of the
// split-K partial traffic.  C == 256, FF % (256 * HS) == 0.
#define FF_HLD 256  // bf16 per row of the hidden image.  No padding (LDS is full: 96 + 15 + 48 KB); instead the 16-byte chunk j of row
                    // r lives at chunk j ^ (r & 31), so the 32 rows of a fragment read hit 32 different chunks
#define FF_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0): this wave's LDS DMA has landed (a barrier does not imply it)
// ABL (debug build, VKN_FFN_ABL; WRONG results, time attribution): 1 = GEMM 2 does not wait for its weight tiles, 2 = no GEMM 2 loop
template <int ABL>
__global__ __launch_bounds__(GM_THREADS, 2) void k_ffn_fused(const float* __restrict__ X, int ldx, const __bf16* __restrict__ W1p,
                                                             const float* __restrict__ b1, const __bf16* __restrict__ W2p, int M,
                                                             int C, int FF, int HS, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_ff[];
    __bf16* Wl = reinterpret_cast<__bf16*>(smem_ff);  // [2][GS_WTILE]
    __bf16* Al = Wl + 2 * GS_WTILE;                   // [2][3][32][40]
    __bf16* Hl = Al + 2 * GS_ATILE;                   // [3][32][FF_HLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.y * GM_BM, hs = blockIdx.x;
    const int kt1 = C >> 5;                    // K-tiles of GEMM 1
    const int cps = (FF >> 8) / HS;            // hidden chunks of this split
    const bool a_role = tid < 256;
    const int ar = (tid >> 3) & 31, aq = tid & 7;
    const size_t aoff = (size_t)min(m0 + ar, M - 1) * ldx + 4 * aq;
    f32x4 ra;

#define FF_DMA(SRC, BUF)                                                                                              \
    do {                                                                                                              \
        const char* gsrc_ = reinterpret_cast<const char*>(SRC) + (size_t)tid * 16;                                    \
        char* ldst_ = reinterpret_cast<char*>(Wl + (size_t)(BUF) * GS_WTILE) + (size_t)wave * 1024;                   \
        _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                                 \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc_ + i * 8192),       \
                                             (__attribute__((address_space(3))) void*)(ldst_ + i * 8192), 16, 0, 0);  \
    } while (0)
#define FF_ASTASH(BUF)                                                                                \
    do {                                                                                              \
        if (a_role) {                                                                                 \
            bf16x4 h_, m_, l_;                                                                        \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
                __bf16 hh_, mm_, ll_;                                                                 \
                vkn_split_bf16x3(ra[e], hh_, mm_, ll_);                                               \
                h_[e] = hh_;                                                                          \
                m_[e] = mm_;                                                                          \
                l_[e] = ll_;                                                                          \
            }                                                                                         \
            __bf16* d_ = Al + (size_t)(BUF) * GS_ATILE + ar * GS_LDR + 4 * aq;                        \
            *reinterpret_cast<bf16x4*>(d_) = h_;                                                      \
            *reinterpret_cast<bf16x4*>(d_ + GM_BM * GS_LDR) = m_;                                     \
            *reinterpret_cast<bf16x4*>(d_ + 2 * GM_BM * GS_LDR) = l_;                                 \
        }                                                                                             \
    } while (0)
#define FF_MFMA6(ACC, AH, AM, AL, BP)                                                                 \
    do {                                                                                              \
        const bf16x8 bh_ = *reinterpret_cast<const bf16x8*>(BP);                                      \
        const bf16x8 bm_ = *reinterpret_cast<const bf16x8*>((BP) + 4 * 256 * 8);                      \
        const bf16x8 bl_ = *reinterpret_cast<const bf16x8*>((BP) + 8 * 256 * 8);                      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bl_, ACC, 0, 0, 0);                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, bh_, ACC, 0, 0, 0);                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, bm_, ACC, 0, 0, 0);                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bm_, ACC, 0, 0, 0);                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, bh_, ACC, 0, 0, 0);                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, bh_, ACC, 0, 0, 0);                         \
    } while (0)

    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    const bool pre_ok = !VKN_ABL_IS(ABL, 1) && !VKN_ABL_IS(ABL, 2) && (kt1 & 1) == 0;   // first tiles of the next GEMM requested one iteration early (they land in buffer 0)

    for (int cc = 0; cc < cps; ++cc) {
        const int c = hs * cps + cc;  // hidden chunk = column tile of W1 = K-tiles 8c .. 8c+7 of W2
        // ---- GEMM 1: acc1 [32 x 256 hidden] = X tile . W1[chunk]^T
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        const __bf16* w1t = W1p + (size_t)c * kt1 * GS_WTILE;
        const __bf16* w2t = W2p + (size_t)c * 8 * GS_WTILE;
        // (chunks after the first: their first W1 tile was requested during the previous chunk's last GEMM-2 iteration)
        if (cc == 0 || !pre_ok) FF_DMA(w1t, 0);
        ra = *reinterpret_cast<const f32x4*>(X + aoff);
        FF_ASTASH(0);
        __syncthreads();
        for (int kt = 0; kt < kt1; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < kt1);
            if (more) {
                FF_DMA(w1t + (size_t)(kt + 1) * GS_WTILE, cur ^ 1);
                ra = *reinterpret_cast<const f32x4*>(X + aoff + (size_t)(kt + 1) * 32);
            } else if (pre_ok) {
                // last K-tile of GEMM 1: buffer 0 was last read one iteration ago (every wave is past that barrier), so GEMM 2's
                // first weight tile can travel under these MFMAs and the hidden-image conversion instead of after them
                FF_DMA(w2t, 0);
            }
            {
                const __bf16* Ab = Al + (size_t)cur * GS_ATILE;
                const __bf16* Wb = Wl + (size_t)cur * GS_WTILE;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const __bf16* ap = Ab + li * GS_LDR + (ks << 4) + (g << 3);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + GM_BM * GS_LDR);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * GM_BM * GS_LDR);
                    const __bf16* bp = Wb + (((ks << 1) + g) * 256 + wave * 32 + li) * 8;
                    FF_MFMA6(acc1, ah, am, al, bp);
                }
            }
            if (more) FF_ASTASH(cur ^ 1);
            __syncthreads();
        }
        // ---- hidden = ReLU(acc1 + b1) -> bf16x3 image [3][32 rows][256 k]; this wave owns hidden columns wave*32 + li
        {
            const float bias = b1[c * 256 + wave * 32 + li];
            const int col = wave * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(acc1[r] + bias, 0.f);
                __bf16 hh, mm, ll;
                vkn_split_bf16x3(v, hh, mm, ll);
                const int row = vkn_cd_row(r, lane);
                __bf16* d = Hl + row * FF_HLD + ((((col >> 3) ^ row) & 31) << 3) + (col & 7);
                d[0] = hh;
                d[GM_BM * FF_HLD] = mm;
                d[2 * GM_BM * FF_HLD] = ll;
            }
        }
        // ---- GEMM 2: acc2 [32 x C] += hidden [32 x 256] . W2[:, chunk]^T   (the barrier below also publishes the hidden image)
        if (!pre_ok) FF_DMA(w2t, 0);
        FF_VMCNT0();
        __syncthreads();
        for (int kt = 0; kt < (VKN_ABL_IS(ABL, 2) ? 0 : 8); ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < 8);
            if (more) FF_DMA(w2t + (size_t)(kt + 1) * GS_WTILE, cur ^ 1);
            else if (pre_ok && cc + 1 < cps)   // ... and the next chunk's first W1 tile under the last GEMM-2 iteration (buffer 0 is free again)
                FF_DMA(W1p + (size_t)(c + 1) * kt1 * GS_WTILE, 0);
            {
                const __bf16* Wb = Wl + (size_t)cur * GS_WTILE;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const __bf16* ap = Hl + li * FF_HLD + (((((kt << 2) + (ks << 1) + g) ^ li) & 31) << 3);
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + GM_BM * FF_HLD);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * GM_BM * FF_HLD);
                    const __bf16* bp = Wb + (((ks << 1) + g) * 256 + wave * 32 + li) * 8;
                    FF_MFMA6(acc2, ah, am, al, bp);
                }
            }
            if (!VKN_ABL_IS(ABL, 1)) FF_VMCNT0();
            __syncthreads();
        }
    }
#undef FF_DMA
#undef FF_ASTASH
#undef FF_MFMA6
#undef FF_VMCNT0

    float* pz = partial + (size_t)hs * M * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + vkn_cd_row(r, lane), col = wave * 32 + li;
        if (row < M && col < C) pz[(size_t)row * C + col] = acc2[r];
    }
}

__global__ void k_rowepi(const float* __restrict__ partial, int ks, int M, int Nout, VknEpi epi);

// out = LN(X + FFN(X)): k_ffn_fused + k_rowepi.  W1s / W2s: pre-split tile images of W1 [FF][C] and W2 [C][FF].
int vkn_launch_ffn_fused(const float* X, int ldx, const void* W1s, const float* b1, const void* W2s, int M, int C, int FF, int HS,
                         float* partial, const VknEpi& epi2, hipStream_t stream) {
    if (C != 256 || FF % (256 * HS) != 0 || HS < 1 || (ldx % 4) != 0) return VKN_E_SHAPE;
    const size_t lds = (size_t)(2 * GS_WTILE + 2 * GS_ATILE + 3 * GM_BM * FF_HLD) * sizeof(__bf16);
#define FFN_LAUNCH(ABLV)                                                                                              \
    do {                                                                                                              \
        VKN_ALLOW_FULL_LDS(k_ffn_fused<ABLV>);                                                                        \
        hipLaunchKernelGGL(k_ffn_fused<ABLV>, dim3(HS, (M + GM_BM - 1) / GM_BM), dim3(GM_THREADS), lds, stream, X, ldx, \
                           static_cast<const __bf16*>(W1s), b1, static_cast<const __bf16*>(W2s), M, C, FF, HS, partial); \
    } while (0)
#ifdef VKN_DEBUG
    const int fabl = vkn_dbg_env("VKN_FFN_ABL", 0);
    if (fabl == 1) FFN_LAUNCH(1);
    else if (fabl == 2) FFN_LAUNCH(2);
    else
#endif
        FFN_LAUNCH(0);
#undef FFN_LAUNCH
    VKN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_rowepi, dim3((M + 3) / 4), dim3(256), 0, stream, partial, HS, M, C, epi2);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// (host) the row epilogue alone: out = epi(sum of `ks` partials [ks][M][Nout]) — the closing step of a z-split GEMM whose result has no
// GEMM consumer (the tracking link's last LayerNorm, vkn_api.hip: run_link_ks)
int vkn_launch_rowepi(const float* partial, int ks, int M, int Nout, const VknEpi& epi, hipStream_t stream) {
    if (!partial || ks < 1 || M <= 0 || Nout <= 0 || Nout > 256) return VKN_E_ARG;
    hipLaunchKernelGGL(k_rowepi, dim3((M + 3) / 4), dim3(256), 0, stream, partial, ks, M, Nout, epi);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// Row epilogue after a split-K GEMM: sums `ks` partials [ks][M][Nout] and applies the epilogue.  One wave per row, Nout <= 256.
__global__ __launch_bounds__(256) void k_rowepi(const float* __restrict__ partial, int ks, int M, int Nout, VknEpi epi) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= M) return;
    VknEpiCols cols;
    vkn_epi_load_cols(epi, Nout, 0, lane, cols);
    VknEpiRow pre;
    vkn_epi_load_row(epi, cols, row, pre);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ks; ++z) {  // fixed order -> deterministic
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += partial[((size_t)z * M + row) * Nout + cols.cidx[q]];
    }
    vkn_row_epilogue(epi, cols, row, Nout, lane, v, pre);
}

// f[row] = ug * LN(params[:, C:2C]; norm_out) + ig * LN(inputf[:, C:2C]; input_norm_out)   (knet/kernel_updator.py:79-88)
__global__ __launch_bounds__(256) void k_ku_mix(const float* __restrict__ params, const float* __restrict__ inputf,
                                                const float* __restrict__ ig, const float* __restrict__ ug,
                                                const float* __restrict__ no_w, const float* __restrict__ no_b,
                                                const float* __restrict__ ino_w, const float* __restrict__ ino_b,
                                                float eps, float* __restrict__ f, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= M) return;
    float po[4], io[4];
    bool ok[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + 64 * q;
        ok[q] = c < C;
        po[q] = ok[q] ? params[(size_t)row * 2 * C + C + c] : 0.f;
        io[q] = ok[q] ? inputf[(size_t)row * 2 * C + C + c] : 0.f;
        s1 += po[q];
        s2 += io[q];
    }
    const float m1 = vkn_wave_sum(s1) / (float)C, m2 = vkn_wave_sum(s2) / (float)C;
    float d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (ok[q]) {
            d1 += (po[q] - m1) * (po[q] - m1);
            d2 += (io[q] - m2) * (io[q] - m2);
        }
    const float r1 = 1.0f / sqrtf(vkn_wave_sum(d1) / (float)C + eps), r2 = 1.0f / sqrtf(vkn_wave_sum(d2) / (float)C + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (ok[q]) {
            const int c = lane + 64 * q;
            const float a = (po[q] - m1) * r1 * no_w[c] + no_b[c];
            const float bb = (io[q] - m2) * r2 * ino_w[c] + ino_b[c];
            f[(size_t)row * C + c] = ug[(size_t)row * C + c] * a + ig[(size_t)row * C + c] * bb;
        }
}

// Scaled-dot-product attention over the kernels of one frame: grid (heads, B, ceil(Nq/16)), 4 waves, one wave per query row.
// q rows: Q[(b*Nq + i)*ldq + h*hd + d]; k/v rows: K[(b*Nk + j)*ldkv + h*hd + d]; out[(b*Nq+i)*ldo + h*hd + d].
// Nk <= 256, hd <= 64, hd % 4 == 0.  The kernel is LDS-bandwidth bound (45 KB of LDS reads per query row with scalar reads),
// so every LDS access is 16 bytes: K/V rows padded to hd+4 floats (conflict-free ds_read_b128), the query row is held in
// registers, the P.V product gives each lane 4 output channels and splits the keys over 64/(hd/4) lane groups.
template <int HD4, int NW = 4>  // hd / 4; waves per workgroup
__global__ __launch_bounds__(64 * NW) void k_attn(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                              const float* __restrict__ Vp, int ldkv, float* __restrict__ out, int ldo,
                                              int Nq, int Nk, float scale, int rpw) {
    // rpw: query rows per workgroup (a multiple of NW: rpw / NW consecutive rows per wave)
    constexpr int hd = HD4 * 4, ldh = hd + 4;
    constexpr int NGRP = 64 / HD4;  // lane groups of the P.V product (HD4 is a power of two <= 16)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* Ks = smem;             // [Nk][hd+4]
    float* Vs = Ks + Nk * ldh;    // [Nk][hd+4]
    float* qs = Vs + Nk * ldh;    // [rpw][hd]  (pre-scaled query rows of this workgroup)
    float* ps = qs + rpw * hd;    // [NW][256]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.z * rpw;
    // K/V staging: one float4 per thread per step, 4 steps batched (8 independent 16-B loads in flight before the LDS writes)
    const int nvec = Nk * HD4;
    constexpr int NT = 64 * NW;
    for (int i0 = tid; i0 < nvec; i0 += NT * 4) {
        f32x4 kv[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * NT, nvec - 1);
            const int j = i / HD4, q4 = i - j * HD4;
            const size_t off = ((size_t)b * Nk + j) * ldkv + h * hd + 4 * q4;
            kv[u] = *reinterpret_cast<const f32x4*>(Kp + off);
            vv[u] = *reinterpret_cast<const f32x4*>(Vp + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT;
            if (i < nvec) {
                const int j = i / HD4, q4 = i - j * HD4;
                *reinterpret_cast<f32x4*>(Ks + j * ldh + 4 * q4) = kv[u];
                *reinterpret_cast<f32x4*>(Vs + j * ldh + 4 * q4) = vv[u];
            }
        }
    }
    for (int i = tid; i < rpw * HD4; i += NT) {
        const int r = i / HD4, q4 = i - r * HD4;
        if (row0 + r < Nq) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Q + ((size_t)b * Nq + row0 + r) * ldq + h * hd + 4 * q4);
            *reinterpret_cast<f32x4*>(qs + r * hd + 4 * q4) = v * scale;
        }
    }
    __syncthreads();
    const int grp = lane / HD4, dl4 = lane - grp * HD4;
    float* myp = ps + wave * 256;
    const int i_begin = row0 + wave * (rpw / NW);
    const int i_end = min(Nq, i_begin + (rpw / NW));
    for (int i = i_begin; i < i_end; ++i) {
        f32x4 qv[HD4];
#pragma unroll
        for (int u = 0; u < HD4; ++u) qv[u] = *reinterpret_cast<const f32x4*>(qs + (i - row0) * hd + 4 * u);  // broadcast reads
        float s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            float a = -INFINITY;
            if (j < Nk) {
                f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < HD4; ++u) acc4 += qv[u] * *reinterpret_cast<const f32x4*>(Ks + j * ldh + 4 * u);
                a = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            }
            s[jj] = a;
            mx = fmaxf(mx, a);
        }
        mx = vkn_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            const float ev = (j < Nk) ? expf(s[jj] - mx) : 0.f;
            s[jj] = ev;
            sum += ev;
        }
        sum = vkn_wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            if (j < Nk) myp[j] = s[jj] * inv;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its own reads
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int j = grp; j < Nk; j += NGRP) o += myp[j] * *reinterpret_cast<const f32x4*>(Vs + j * ldh + 4 * dl4);
#pragma unroll
        for (int off = HD4; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = o[e];
                o[e] = t + __shfl_xor(t, off, 64);
            }
        }
        if (grp == 0) *reinterpret_cast<f32x4*>(out + ((size_t)b * Nq + i) * ldo + h * hd + 4 * dl4) = o;
        __builtin_amdgcn_wave_barrier();
    }
}

// Attention on the matrix cores (north_star: "MFMA only for the dense N x C kernel-attention contraction"): one workgroup per
// (frame, head), one wave per block of 32 queries; both contractions are 32x32x16 bf16 MFMAs on operands split into THREE bf16
// planes (hi + mid + lo = the fp32 value to 24 bits) with the six cross products of the [N x C] GEMMs (hi lo, lo hi, mid mid,
// hi mid, mid hi, hi hi; smallest first) — fp32-class accuracy.  (A first version used the f16 hi + lo split of the gather / decode
// kernels, three products: ~2^-21 per term, 1e-5 on the output — 20x the error of an fp32 loop and enough to move a near-threshold
// pixel of the free-running head; the bf16x3 form costs 2x the MFMAs of a kernel that is latency-bound.)
//   S^T = K . Q^T      A = K rows from LDS (three planes), B = the wave's 32 scaled queries, split in registers.  The accumulator
//                      layout puts a QUERY on a lane column, so the softmax over the keys is a reduction over a lane's own registers
//                      plus one exchange with lane ^ 32 — no row-wise cross-lane reductions;
//   O   = P . V        A = P: the normalised probabilities ARE already an A fragment (row = query = lane column) if the contraction
//                      index walks the keys in the order the accumulator holds them; B = V^T from LDS, stored transposed with exactly
//                      that key permutation (position of key 32 kb + jj: 32 kb + 16 (jj >> 4) + 8 ((jj >> 2) & 1) + 4 ((jj >> 3) & 1)
//                      + (jj & 3)), so a B fragment is one 16-byte LDS read.  Output rows = queries, lane = channel: coalesced stores.
// HD: head width (16 / 32 / 64), NKB: 32-key blocks (keys >= Nk are masked to -inf / zero).  Replaces k_attn's VALU loops
// (one wave per query row: 28-31 us per launch at 32 frames).
#define ATTM_MFMA6(ACC, AH, AM, AL, BH, BM, BL)                                   \
    do {                                                                          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM, ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM, ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, ACC, 0, 0, 0);      \
    } while (0)
template <int HD, int NKB>
__global__ __launch_bounds__(256) void k_attn_mfma(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                   const float* __restrict__ Vp, int ldkv, float* __restrict__ out, int ldo,
                                                   int Nq, int Nk, float scale) {
    constexpr int NKP = NKB * 32;            // padded key count
    constexpr int KLD = HD + 8;              // bf16 per K row (16-byte aligned rows, conflict-free 16-B reads)
    constexpr int VLD = NKP + 8;             // bf16 per V^T row
    constexpr int DP = HD < 32 ? 32 : HD;    // channel rows of V^T (one or two 32-column blocks of O)
    constexpr int NDB = DP / 32;
    constexpr int KS = HD / 16;              // k-steps of K . Q^T
    extern __shared__ __attribute__((aligned(16))) char smem_am[];
    __bf16* Kpl = reinterpret_cast<__bf16*>(smem_am);      // [3][NKP][KLD]: hi, mid, lo planes of the K rows
    __bf16* Vpl = Kpl + 3 * NKP * KLD;                      // [3][DP][VLD]: planes of V^T, keys permuted
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    // ---- every global load of the prologue is requested FIRST (K rows, V columns, the wave's queries): one round trip instead of
    // one per staging-loop iteration (round 5: the loops ran load -> wait -> write eight times over, 8 of the kernel's 11 us)
    constexpr int ITEMS = NKP * (HD / 4);    // staging items: (key, 4 channels) of K = (4 keys, channel) of V^T
    constexpr int NIT = (ITEMS + 255) / 256; // ... per thread
    f32x4 kreg[NIT];
    float vreg[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(tid + 256 * it, ITEMS - 1);
        const int j = i / (HD / 4), d4 = i - j * (HD / 4);
        kreg[it] = *reinterpret_cast<const f32x4*>(Kp + ((size_t)b * Nk + min(j, Nk - 1)) * ldkv + h * HD + 4 * d4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(tid + 256 * it, ITEMS - 1);
        const int j4 = i / HD, d = i - j4 * HD;
#pragma unroll
        for (int e = 0; e < 4; ++e) vreg[it][e] = Vp[((size_t)b * Nk + min(4 * j4 + e, Nk - 1)) * ldkv + h * HD + d];
    }
    const int qb = blockIdx.z * 4 + wave;
    f32x4 qreg[KS][2];
    {
        const int q = min(qb * 32 + li, Nq - 1);
        const float* qp = Q + ((size_t)b * Nq + q) * ldq + h * HD + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qreg[ks][0] = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
            qreg[ks][1] = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
        }
    }
    // ---- stage K (rows) as three bf16 planes: a thread owns 4 channels of a key
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        if (ITEMS % 256 != 0 && i >= ITEMS) break;
        const int j = i / (HD / 4), d4 = i - j * (HD / 4);
        const f32x4 kv = (j < Nk) ? kreg[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x4 kh, km, kl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 hh, mm, ll;
            vkn_split_bf16x3(kv[e], hh, mm, ll);
            kh[e] = hh;
            km[e] = mm;
            kl[e] = ll;
        }
        __bf16* kd = Kpl + j * KLD + 4 * d4;
        *reinterpret_cast<bf16x4*>(kd) = kh;
        *reinterpret_cast<bf16x4*>(kd + NKP * KLD) = km;
        *reinterpret_cast<bf16x4*>(kd + 2 * NKP * KLD) = kl;
    }
    // ---- ... and V transposed with the keys permuted: a thread owns ONE channel of 4 consecutive keys — the permutation keeps
    // such a group contiguous, so each plane takes one 8-byte store (lanes = channels: the four row loads are coalesced)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        if (ITEMS % 256 != 0 && i >= ITEMS) break;
        const int j4 = i / HD, d = i - j4 * HD;
        const int j = 4 * j4, jj = j & 31;
        const int pos = (j & ~31) + 16 * (jj >> 4) + 8 * ((jj >> 2) & 1) + 4 * ((jj >> 3) & 1);
        bf16x4 vh, vm, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (j + e < Nk) ? vreg[it][e] : 0.f;
            __bf16 hh, mm, ll;
            vkn_split_bf16x3(v, hh, mm, ll);
            vh[e] = hh;
            vm[e] = mm;
            vl[e] = ll;
        }
        __bf16* vd = Vpl + d * VLD + pos;
        *reinterpret_cast<bf16x4*>(vd) = vh;
        *reinterpret_cast<bf16x4*>(vd + DP * VLD) = vm;
        *reinterpret_cast<bf16x4*>(vd + 2 * DP * VLD) = vl;
    }
    if (HD < 32)   // channel rows HD .. 31 of V^T: zeros (they feed output columns nobody stores, but must be finite)
        for (int i = tid; i < 3 * (DP - HD) * NKP; i += 256) {
            const int pl = i / ((DP - HD) * NKP), r = i - pl * (DP - HD) * NKP;
            Vpl[(pl * DP + HD + r / NKP) * VLD + (r - (r / NKP) * NKP)] = (__bf16)0.f;
        }
    __syncthreads();
    if (qb * 32 >= Nq) return;   // (no barrier below)
    // ---- the wave's 32 queries as B fragments: lane (g, li) = query li, channels 16 ks + 8 g .. + 7, scaled, split
    bf16x8 qh[KS], qm[KS], ql[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 hh, mm, ll;
            vkn_split_bf16x3((e < 4 ? qreg[ks][0][e] : qreg[ks][1][e - 4]) * scale, hh, mm, ll);
            qh[ks][e] = hh;
            qm[ks][e] = mm;
            ql[ks][e] = ll;
        }
    }
    // ---- S^T = K . Q^T: acc[kb][r] at lane (g, li) = score of key 32 kb + cd_row(r, lane) for query li
    f32x16 acc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const __bf16* kp = Kpl + (kb * 32 + li) * KLD + 16 * ks + 8 * g;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(kp);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(kp + NKP * KLD);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(kp + 2 * NKP * KLD);
            ATTM_MFMA6(acc[kb], ah, am, al, qh[ks], qm[ks], ql[ks]);
        }
    }
    // ---- softmax over the keys of this lane's query (its own registers + the other half-wave)
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (kb * 32 + vkn_cd_row(r, lane)) < Nk;
            acc[kb][r] = ok ? acc[kb][r] : -INFINITY;
            mx = fmaxf(mx, acc[kb][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ev = __expf(acc[kb][r] - mx);   // (-inf -> 0)  v_exp_f32 on x log2(e): exponent error <= |x| 9e-8, i.e. < 1 ulp on every term that is
                                                          // not negligible against the row maximum; expf's ~12 instructions x 64 per lane were 1 us of this kernel
            acc[kb][r] = ev;
            sum += ev;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // ---- O = P . V: k-step t = (kb, half): the lane's registers 8 half .. 8 half + 7 of acc[kb] ARE the A fragment
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bf16x8 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 hh, mm, ll;
                vkn_split_bf16x3(acc[kb][8 * hf + e] * inv, hh, mm, ll);
                ph[e] = hh;
                pm[e] = mm;
                pl[e] = ll;
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const __bf16* vp = Vpl + (db * 32 + li) * VLD + kb * 32 + 16 * hf + 8 * g;
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vp);
                const bf16x8 vm = *reinterpret_cast<const bf16x8*>(vp + DP * VLD);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vp + 2 * DP * VLD);
                ATTM_MFMA6(o[db], ph, pm, pl, vh, vm, vl);
            }
        }
    // ---- o[db][r] at lane (g, li) = output of query 32 qb + cd_row(r, lane), channel 32 db + li
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const int d = db * 32 + li;
        if (d < HD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + vkn_cd_row(r, lane);
                if (q < Nq) out[((size_t)b * Nq + q) * ldo + h * HD + d] = o[db][r];
            }
        }
    }
}
#undef ATTM_MFMA6

// Attention with MORE than 256 keys per query (the clip-level query merge of the VIS heads: Nk = frames * kernels, a query row
// attends to every frame's kernels — knet_vis/tracker/kernel_frame_iter_head.py:142-160).  One wave per query row, grid
// (heads, B, ceil(Nq / 4)); K / V stay in global memory (B * Nk * hd floats per head: cache-resident), the row's scores live in
// LDS ([4][Nk] floats).  Same lane roles as k_attn: a lane owns keys lane, lane + 64, .. for q.K and 4 output channels of a key
// group for P.V.  Runs once per clip, not per stage — not a hot kernel.
template <int HD4>
__global__ __launch_bounds__(256) void k_attn_long(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                   const float* __restrict__ Vp, int ldkv, float* __restrict__ out, int ldo,
                                                   int Nq, int Nk, float scale) {
    constexpr int hd = HD4 * 4;
    constexpr int NGRP = 64 / HD4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = reinterpret_cast<float*>(smem_raw) + (size_t)wave * Nk;
    const int i = blockIdx.z * 4 + wave;
    if (i >= Nq) return;  // (no workgroup-level barrier below: a wave only reads its own LDS rows)
    f32x4 qv[HD4];
#pragma unroll
    for (int u = 0; u < HD4; ++u) qv[u] = *reinterpret_cast<const f32x4*>(Q + ((size_t)b * Nq + i) * ldq + h * hd + 4 * u) * scale;
    const float* Kb = Kp + (size_t)b * Nk * ldkv + h * hd;
    const float* Vb = Vp + (size_t)b * Nk * ldkv + h * hd;
    float mx = -INFINITY;
    for (int j = lane; j < Nk; j += 64) {
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < HD4; ++u) acc4 += qv[u] * *reinterpret_cast<const f32x4*>(Kb + (size_t)j * ldkv + 4 * u);
        const float a = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    mx = vkn_wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Nk; j += 64) {  // (each lane re-reads only what it wrote)
        const float ev = expf(sc[j] - mx);
        sc[j] = ev;
        sum += ev;
    }
    sum = vkn_wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int grp = lane / HD4, dl4 = lane - grp * HD4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int j = grp; j < Nk; j += NGRP) o += (sc[j] * inv) * *reinterpret_cast<const f32x4*>(Vb + (size_t)j * ldkv + 4 * dl4);
#pragma unroll
    for (int off = HD4; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = o[e];
            o[e] = t + __shfl_xor(t, off, 64);
        }
    }
    if (grp == 0) *reinterpret_cast<f32x4*>(out + ((size_t)b * Nq + i) * ldo + h * hd + 4 * dl4) = o;
}

// out[r][c] = a[r][c] + pos[r % period][c]   (query_pos / key_pos of the 'attention_pos' query merge: one [Np][C] table, repeated
// per clip and per frame — kernel_frame_iter_head.py:154-155); C % 4 == 0
__global__ __launch_bounds__(256) void k_add_rows(const float* __restrict__ a, const float* __restrict__ pos,
                                                  float* __restrict__ out, size_t rows, int C4, int period) {
    const size_t n = rows * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] +
                                           reinterpret_cast<const f32x4*>(pos)[(r % period) * C4 + c];
    }
}

int vkn_launch_add_rows(const float* a, const float* pos, float* out, size_t rows, int C, int period, hipStream_t st) {
    if ((C & 3) || period <= 0) return VKN_E_SHAPE;
    if (rows == 0) return VKN_OK;
    const size_t n = rows * (C / 4);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_add_rows, dim3(grid), dim3(256), 0, st, a, pos, out, rows, C / 4, period);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// Bilinear upsample by integer factor S, align_corners=False (F.interpolate(scale_factor=S, mode='bilinear')):
// src = (dst + 0.5) / S - 0.5 clamped at 0; neighbours clamped at the border.
// Write-bound (S*S outputs per input): one workgroup = one input row of one plane -> its S output rows; a thread owns 4
// consecutive output x (x-weights computed once, reused for the S rows) and streams them out with 16-B non-temporal stores.
template <int NT, int ROWS>  // NT: non-temporal stores; ROWS: input rows per workgroup
__global__ __launch_bounds__(256) void k_upsample(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                  int S) {
    const int OW = W * S, OH = H * S;
    const int plane = blockIdx.y;
    const float rs = 1.0f / (float)S;
    const float* ip = in + (size_t)plane * H * W;
    const bool vec = ((OW & 3) == 0);
    for (int ox4 = threadIdx.x * 4; ox4 < OW; ox4 += 1024) {
        int x0[4], x1[4];
        float lx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sx = fmaxf(((float)(ox4 + k) + 0.5f) * rs - 0.5f, 0.f);
            x0[k] = min((int)sx, W - 1);
            x1[k] = min(x0[k] + 1, W - 1);
            lx[k] = sx - (float)x0[k];
        }
        for (int yy = 0; yy < ROWS; ++yy) {
            const int y = blockIdx.x * ROWS + yy;
            if (y >= H) break;
            for (int j = 0; j < S; ++j) {
                const int oy = y * S + j;
                const float sy = fmaxf(((float)oy + 0.5f) * rs - 0.5f, 0.f);
                const int y0 = (int)sy;
                const int y1 = min(y0 + 1, H - 1);
                const float ly = sy - (float)y0, hy = 1.f - ly;
                const float* r0 = ip + (size_t)y0 * W;
                const float* r1 = ip + (size_t)y1 * W;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float hx = 1.f - lx[k];
                    o[k] = hy * (hx * r0[x0[k]] + lx[k] * r0[x1[k]]) + ly * (hx * r1[x0[k]] + lx[k] * r1[x1[k]]);
                }
                float* op = out + ((size_t)plane * OH + oy) * OW + ox4;
                if (vec && ox4 + 3 < OW) {
                    if (NT)
                        __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(op));
                    else
                        *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
                } else {
                    for (int k = 0; k < 4 && ox4 + k < OW; ++k) op[k] = o[k];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host launchers
static int gemm_check(const VknGemmProb& p, int M, int K, int ksplit, const float* partial) {
    if (M <= 0 || K <= 0 || p.Nout <= 0 || K % GM_KT != 0) return VKN_E_SHAPE;
    // LayerNorm / dot side output need the whole (sub-)row in one 256-column tile
    if (p.epi.dot_vec && p.Nout > GM_BN) return VKN_E_SHAPE;
    if (p.epi.ln_w && (p.Nout - p.epi.ln_from_col > GM_BN || p.epi.ln_from_col % GM_BN != 0)) return VKN_E_SHAPE;
    if (ksplit > 1 && (p.Nout > GM_BN || !partial)) return VKN_E_SHAPE;
    return VKN_OK;
}

static int gemm_one_exact(const VknGemmProb& p, int M, int K, int ksplit, float* partial, hipStream_t stream) {
    dim3 grid((p.Nout + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, ksplit);
    hipLaunchKernelGGL(k_gemm, grid, dim3(GM_THREADS), 0, stream, p.A, p.A2, p.A3, p.A4, p.lda, p.W, M, K, p.Nout, partial, p.epi);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// One or two independent GEMMs (same M, K).  Split (bf16x3) path: a single launch; exact path: one launch each.
int vkn_launch_gemm_group(const VknGemmProb* probs, int nprob, int M, int K, int ksplit, float* partial,
                          hipStream_t stream) {
    if (nprob < 1 || nprob > 2) return VKN_E_ARG;
    if (ksplit < 1) ksplit = 1;
    if (nprob > 1 && ksplit > 1) return VKN_E_ARG;
    bool split = true;
    for (int i = 0; i < nprob; ++i) {
        const int rc = gemm_check(probs[i], M, K, ksplit, partial);
        if (rc != VKN_OK) return rc;
        split = split && probs[i].Wsplit && (probs[i].lda % 4) == 0;
    }
    if (split && (K == 256 || K == 512 || K == 768) && ksplit == 1 && vkn_dbg_env("VKN_GEMM_T3", 1) != 0) {
        // the register-streaming kernel (vkn_chain.hip): resident A image, no barrier in the K loop — ~0.5 us per K-tile where the
        // LDS-DMA loop below costs 0.87 (13.8 -> 9 us per launch at 117 rows)
        const int rc = vkn_launch_gemm_t3(probs, nprob, M, K, stream);
        if (rc != VKN_E_SHAPE) return rc;
    }
    if (split) {
        const size_t lds = (size_t)(3 * GS_WTILE + 2 * GS_ATILE) * sizeof(__bf16);  // 162,816 B of the 163,840
        int nmax = probs[0].Nout;
        if (nprob > 1 && probs[1].Nout > nmax) nmax = probs[1].Nout;
        dim3 grid((nmax + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, nprob > 1 ? nprob : ksplit);
#define GS3_LAUNCH(ABLV)                                                                                                  \
    do {                                                                                                                  \
        VKN_ALLOW_FULL_LDS((k_gemm_s3<ABLV>));                                                                            \
        hipLaunchKernelGGL((k_gemm_s3<ABLV>), grid, dim3(GM_THREADS), lds, stream, probs[0], probs[nprob > 1 ? 1 : 0], nprob, M, K, \
                           partial);                                                                                      \
    } while (0)
#ifdef VKN_DEBUG
#define GR3_LAUNCH(ABLV)                                                                                                  \
    do {                                                                                                                  \
        const size_t lds_r3 = (size_t)2 * GS_ATILE * sizeof(__bf16) + (size_t)GM_BM * GM_LDT * sizeof(float);              \
        hipLaunchKernelGGL((k_gemm_r3<ABLV>), grid, dim3(GM_THREADS), lds_r3, stream, probs[0], probs[nprob > 1 ? 1 : 0], nprob, M, \
                           K, partial);                                                                                   \
    } while (0)
        const bool r3 = vkn_dbg_env("VKN_GEMM_R3", 0) != 0;  // A/B: register-streamed weights (k_gemm_r3)
        if (vkn_dbg_env("VKN_GEMM_X16", 0) != 0 && ksplit == 1) {  // A/B: 16 waves, intra-workgroup split-K (k_gemm_x16)
            const size_t lds_x = (size_t)4 * GS_ATILE * sizeof(__bf16) + (size_t)2 * GM_BM * GM_LDT * sizeof(float);
            const int xabl = vkn_dbg_env("VKN_GEMM_ABL", 0);
            VKN_ALLOW_FULL_LDS((k_gemm_x16<0>));
            VKN_ALLOW_FULL_LDS((k_gemm_x16<1>));
            if (xabl == 1) hipLaunchKernelGGL((k_gemm_x16<1>), grid, dim3(1024), lds_x, stream, probs[0], probs[nprob > 1 ? 1 : 0], nprob, M, K);
            else hipLaunchKernelGGL((k_gemm_x16<0>), grid, dim3(1024), lds_x, stream, probs[0], probs[nprob > 1 ? 1 : 0], nprob, M, K);
        } else {
        const int gabl = vkn_dbg_env("VKN_GEMM_ABL", 0);
        if (gabl == 1) { if (r3) GR3_LAUNCH(1); else GS3_LAUNCH(1); }
        else if (gabl == 2) { if (r3) GR3_LAUNCH(2); else GS3_LAUNCH(2); }
        else if (gabl == 3) GR3_LAUNCH(3);
        else if (gabl == 4) GR3_LAUNCH(4);
        else if (gabl == 5) GR3_LAUNCH(5);
        else if (gabl == 6) GR3_LAUNCH(6);
        else if (r3) GR3_LAUNCH(0);
        else GS3_LAUNCH(0);
        }
#else
        GS3_LAUNCH(0);
#endif
#undef GS3_LAUNCH
#ifdef VKN_DEBUG
#undef GR3_LAUNCH
#endif
        VKN_CHECK_LAUNCH();
    } else {
        for (int i = 0; i < nprob; ++i) {
            const int rc = gemm_one_exact(probs[i], M, K, ksplit, partial, stream);
            if (rc != VKN_OK) return rc;
        }
    }
    if (ksplit > 1) {
        hipLaunchKernelGGL(k_rowepi, dim3((M + 3) / 4), dim3(256), 0, stream, partial, ksplit, M, probs[0].Nout, probs[0].epi);
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}

int vkn_launch_gemm(const float* A, const float* A2, int lda, const float* W, const void* Wsplit, int M, int K, int Nout,
                    int ksplit, float* partial, const VknEpi& epi, hipStream_t stream) {
    VknGemmProb p{A, A2, nullptr, nullptr, lda, W, Wsplit, Nout, epi};
    return vkn_launch_gemm_group(&p, 1, M, K, ksplit, partial, stream);
}

// bytes of the pre-split tile images of a [Nout][K] weight
size_t vkn_split_w3_bytes(int Nout, int K) { return (size_t)((Nout + 255) / 256) * (K / 32) * GS_WTILE * sizeof(__bf16); }

__global__ void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Cc) return;
    const int r = i / Cc, c = i - r * Cc;
    dst[(size_t)c * R + r] = src[i];
}

int vkn_launch_transpose(const float* src, float* dst, int R, int Cc, hipStream_t st) {
    const int n = R * Cc;
    hipLaunchKernelGGL(k_transpose, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, R, Cc);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// fp32 W [Nout][K] -> tile images (K % 32 == 0)
int vkn_launch_split_w3(const float* W, void* Wp, int Nout, int K, hipStream_t stream) {
    if (Nout <= 0 || K <= 0 || K % 32 != 0) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_split_w3, dim3(K / 32, (Nout + 255) / 256), dim3(256), 0, stream, W, static_cast<__bf16*>(Wp), Nout, K,
                       (size_t)K, (size_t)1);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// tile images of the TRANSPOSE of a stored matrix: Wt [Nout][K] with Wt[n][k] = W[k * Nout + n] (W is [K][Nout] row-major)
int vkn_launch_split_w3_t(const float* W, void* Wp, int Nout, int K, hipStream_t stream) {
    if (Nout <= 0 || K <= 0 || K % 32 != 0) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_split_w3, dim3(K / 32, (Nout + 255) / 256), dim3(256), 0, stream, W, static_cast<__bf16*>(Wp), Nout, K,
                       (size_t)1, (size_t)Nout);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_ku_mix(const float* params, const float* inputf, const float* ig, const float* ug, const float* no_w,
                      const float* no_b, const float* ino_w, const float* ino_b, float eps, float* f, int M, int C,
                      hipStream_t stream) {
    if (C > 256) return VKN_E_SHAPE;
    hipLaunchKernelGGL(k_ku_mix, dim3((M + 3) / 4), dim3(256), 0, stream, params, inputf, ig, ug, no_w, no_b, ino_w, ino_b,
                       eps, f, M, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* out, int ldo, int B, int Nq,
                    int Nk, int heads, int hd, hipStream_t stream) {
    if (hd > 64 || hd < 4 || (hd & (hd - 1)) != 0 || (ldq % 4) || (ldkv % 4) || (ldo % 4)) return VKN_E_SHAPE;
    if (Nk > 256) {  // clip-level query merge: scores in LDS, K / V from global
        const size_t ldsl = (size_t)4 * Nk * sizeof(float);
        if (ldsl > 160 * 1024) return VKN_E_SHAPE;
        dim3 gl(heads, B, (Nq + 3) / 4);
        const float scl = 1.0f / sqrtf((float)hd);
#define ATTL_CASE(H4)                                                                                                     \
    case H4:                                                                                                              \
        if (ldsl > 64 * 1024) VKN_ALLOW_FULL_LDS(k_attn_long<H4>);                                                        \
        hipLaunchKernelGGL(k_attn_long<H4>, gl, dim3(256), ldsl, stream, Q, ldq, K, V, ldkv, out, ldo, Nq, Nk, scl);      \
        break;
        switch (hd / 4) {
            ATTL_CASE(1) ATTL_CASE(2) ATTL_CASE(4) ATTL_CASE(8) ATTL_CASE(16)
            default: return VKN_E_SHAPE;
        }
#undef ATTL_CASE
        VKN_CHECK_LAUNCH();
        return VKN_OK;
    }
    // matrix-core attention: head widths 16 / 32 with up to 256 keys, 64 with up to 128 (three bf16 planes of K and V^T in <= 160 KB)
    if ((hd == 16 || hd == 32 || (hd == 64 && Nk <= 128)) && Nk <= 256 && !vkn_dbg_env("VKN_ATTN_VALU", 0)) {
        const int nkb = (Nk + 31) / 32;
        const int nkbt = nkb <= 1 ? 1 : nkb <= 2 ? 2 : nkb <= 4 ? 4 : nkb <= 6 ? 6 : 8;
        const int dp = hd < 32 ? 32 : hd;
        const size_t ldsm = ((size_t)3 * nkbt * 32 * (hd + 8) + (size_t)3 * dp * (nkbt * 32 + 8)) * sizeof(__bf16);   // three planes of K and V^T
        dim3 gm(heads, B, (Nq + 127) / 128);
        const float sc = 1.0f / sqrtf((float)hd);
#define ATTM_LAUNCH(HDV, NKBV)                                                                                            \
    do {                                                                                                                  \
        if (ldsm > 64 * 1024) VKN_ALLOW_FULL_LDS((k_attn_mfma<HDV, NKBV>));                                               \
        hipLaunchKernelGGL((k_attn_mfma<HDV, NKBV>), gm, dim3(256), ldsm, stream, Q, ldq, K, V, ldkv, out, ldo, Nq, Nk, sc); \
    } while (0)
#define ATTM_HD(HDV)                                       \
    do {                                                   \
        switch (nkbt) {                                    \
            case 1: ATTM_LAUNCH(HDV, 1); break;            \
            case 2: ATTM_LAUNCH(HDV, 2); break;            \
            case 4: ATTM_LAUNCH(HDV, 4); break;            \
            case 6: ATTM_LAUNCH(HDV, 6); break;            \
            default: ATTM_LAUNCH(HDV, 8); break;           \
        }                                                  \
    } while (0)
        if (hd == 16) ATTM_HD(16);
        else if (hd == 32) ATTM_HD(32);
        else ATTM_HD(64);
#undef ATTM_HD
#undef ATTM_LAUNCH
        VKN_CHECK_LAUNCH();
        return VKN_OK;
    }
    // (head widths 4 / 8: the VALU kernel)
    // 16 query rows per workgroup.  64 (K / V of a (frame, head) staged twice instead of eight times) was measured at 32 frames
    // per call: 36 us instead of 30 — the kernel is bound by the serial per-row chain of a wave, not by the staging
    const int rpw = 16;
    // (8 waves x 2 rows instead of 4 x 4 — template parameter NW — was measured: 28-31 us at 32 frames either way)
    const int nw = 4;
    const size_t lds = ((size_t)2 * Nk * (hd + 4) + (size_t)rpw * hd + (size_t)nw * 256) * sizeof(float);
    if (lds > 160 * 1024) return VKN_E_SHAPE;   // (N = 216 kernels of 32-wide heads: 68 KB — beyond the 64 KB default limit)
    dim3 grid(heads, B, (Nq + rpw - 1) / rpw);
    const float scale = 1.0f / sqrtf((float)hd);
#define ATT_CASE(H4)                                                                                                      \
    case H4:                                                                                                              \
        if (lds > 64 * 1024) VKN_ALLOW_FULL_LDS((k_attn<H4, 4>));                                                         \
        hipLaunchKernelGGL((k_attn<H4, 4>), grid, dim3(256), lds, stream, Q, ldq, K, V, ldkv, out, ldo, Nq, Nk, scale, rpw); \
        break;
    switch (hd / 4) {
        ATT_CASE(1) ATT_CASE(2) ATT_CASE(4) ATT_CASE(8) ATT_CASE(16)
        default: return VKN_E_SHAPE;
    }
#undef ATT_CASE
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// Register-staged variant for S = 2 and S = 4 (every shipped config): the generic kernel issues 16 gather loads per 16-byte store
// and reaches 4.6 TB/s of writes where a plain fill reaches 6.9 TB/s (tools/write_ceiling.py).  Here a thread owns one quad of
// 4 consecutive output columns for UP_ROWS input rows: it loads the UP_ROWS + 2 input rows x (4 / S + 2) input columns its
// quad can touch ONCE (coalesced across the workgroup), interpolates them horizontally once, and then only blends two of
// those values per output pixel.  Coefficients come from the same clamped source-index formula as the generic kernel, taps are
// selected from the staged registers -> the same values enter the same expression.
#define UP_ROWS 4
// O16 = 1 (opt-in, VKN_FLAG_SCALED_F16 / vkn_upsample_bilinear_f16out): the up-scaled logits leave as fp16 — the SAME fp32 arithmetic,
// one round-to-nearest at the store (8-byte stores per quad): half of the 245 MB per frame this kernel writes, which is 45 % of a step
template <int S, int NT, int SUBS, int O16 = 0>  // SUBS consecutive groups of UP_ROWS input rows per workgroup (contiguous S*UP_ROWS*SUBS output rows)
__global__ __launch_bounds__(256) void k_upsample_s(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
    constexpr int NTAP = 4 / S + 2;  // input columns a quad can touch: jb - 1 .. jb + 4 / S
    const int OW = W * S, OH = H * S;
    const int plane = blockIdx.y;
    const int yb0 = blockIdx.x * UP_ROWS * SUBS;  // first input row of this workgroup
    const float rs = 1.0f / (float)S;
    const float* ip = in + (size_t)plane * H * W;
    float* op = out + (size_t)plane * OH * OW;
    _Float16* op16 = reinterpret_cast<_Float16*>(out) + (size_t)plane * OH * OW;   // (O16: `out` points at 2-byte elements)
    // NT == 5 (debug A/B, round 4; measured SLOWER: 1555 vs 1470 us per 32-frame launch): the workgroup's UP_ROWS * SUBS + 2 input rows
    // go through LDS once — 18 load instructions per thread instead of 66 (three overlapping tap loads per row) — and the taps come
    // from LDS.  Fewer vector-memory instructions do not help: the staging phase + barrier in front of a workgroup's first store costs
    // more than the tap loads that ride between the store bursts.  Slot r holds input row clamp(yb0 - 1 + r).
    extern __shared__ __attribute__((aligned(16))) float up_rows[];
    if (VKN_ABL_IS(NT, 5)) {
        for (int idx = threadIdx.x; idx < (UP_ROWS * SUBS + 2) * W; idx += 256) {
            const int r = idx / W, c = idx - r * W;
            up_rows[idx] = ip[(size_t)min(max(yb0 - 1 + r, 0), H - 1) * W + c];
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q * 4 < OW; q += 256) {
        const int jb = (q * 4) / S;  // first input column under the quad
        // horizontal coefficients of the quad's 4 output columns, local tap indices
        int t0[4], t1[4], cx[NTAP];
        float lx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sx = fmaxf(((float)(q * 4 + k) + 0.5f) * rs - 0.5f, 0.f);
            const int x0 = min((int)sx, W - 1), x1 = min(x0 + 1, W - 1);
            lx[k] = sx - (float)x0;
            t0[k] = x0 - (jb - 1);
            t1[k] = x1 - (jb - 1);
        }
#pragma unroll
        for (int i = 0; i < NTAP; ++i) cx[i] = min(max(jb - 1 + i, 0), W - 1);
        auto hinterp = [&](const float (&v)[NTAP], float (&h)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = v[0], b2 = v[0];
#pragma unroll
                for (int i = 1; i < NTAP; ++i) {
                    a = (t0[k] == i) ? v[i] : a;
                    b2 = (t1[k] == i) ? v[i] : b2;
                }
                h[k] = (1.f - lx[k]) * a + lx[k] * b2;
            }
        };
        // rows yb - 1 .. yb + UP_ROWS (clamped) of the current group, interpolated horizontally: hrow[r][k]
        float hrow[UP_ROWS + 2][4];
#pragma unroll
        for (int r = 0; r < UP_ROWS + 2; ++r) {
            const int y = min(max(yb0 - 1 + r, 0), H - 1);
            float v[NTAP];
#pragma unroll
            for (int i = 0; i < NTAP; ++i) v[i] = VKN_ABL_IS(NT, 5) ? up_rows[r * W + cx[i]] : VKN_ABL_IS(NT, 2) ? (float)(y + cx[i]) : VKN_ABL_IS(NT, 3) ? __builtin_nontemporal_load(ip + (size_t)y * W + cx[i]) : ip[(size_t)y * W + cx[i]];  // NT == 2: write-only ablation, 3: nontemporal input loads (debug A/B)
            hinterp(v, hrow[r]);
        }
        // NT == 4: ALL input rows of the workgroup are requested up front — no load sits between store bursts
        float pre[NT == 4 ? SUBS : 1][UP_ROWS][NTAP];
        if (NT == 4) {
#pragma unroll
            for (int sb = 0; sb + 1 < SUBS; ++sb)
#pragma unroll
                for (int r = 0; r < UP_ROWS; ++r) {
                    const int y = min(yb0 + sb * UP_ROWS + UP_ROWS + 1 + r, H - 1);
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) pre[sb][r][i] = ip[(size_t)y * W + cx[i]];
                }
        }
        // (unrolled on purpose: the SUBS copies keep their 16 rows' offsets and blend weights in SGPRs at once — 498 SGPR spills as
        //  v_writelane / v_readlane — but `#pragma unroll 1` (33 spills, 65 instead of 200 VGPRs) was measured 4 % SLOWER: 1.53-1.54 ms
        //  against 1.48 ms per 32-frame launch; the spill code is scalar work beside a store-bound stream)
#pragma unroll
        for (int sub = 0; sub < SUBS; ++sub) {
            const int yb = yb0 + sub * UP_ROWS;
            if (yb >= H) break;
            // the next group's UP_ROWS new input rows are requested BEFORE this group's 16 stores: vector memory operations
            // retire in order, so loads issued behind a store burst would wait for it
            float nv[UP_ROWS][NTAP];
            if (NT == 4) {
                if (sub + 1 < SUBS) {
#pragma unroll
                    for (int r = 0; r < UP_ROWS; ++r)
#pragma unroll
                        for (int i = 0; i < NTAP; ++i) nv[r][i] = pre[sub][r][i];
                }
            } else if (sub + 1 < SUBS) {
#pragma unroll
                for (int r = 0; r < UP_ROWS; ++r) {
                    const int y = min(yb + UP_ROWS + 1 + r, H - 1);
#pragma unroll
                    for (int i = 0; i < NTAP; ++i) nv[r][i] = VKN_ABL_IS(NT, 5) ? up_rows[(sub * UP_ROWS + UP_ROWS + 2 + r) * W + cx[i]] : VKN_ABL_IS(NT, 2) ? (float)(y + cx[i]) : VKN_ABL_IS(NT, 3) ? __builtin_nontemporal_load(ip + (size_t)y * W + cx[i]) : ip[(size_t)y * W + cx[i]];
                }
            }
            // vertical blend + store: output rows S * yb .. S * (yb + UP_ROWS) - 1
#pragma unroll
            for (int i = 0; i < UP_ROWS; ++i) {
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const int oy = (yb + i) * S + j;
                    if (oy >= OH) continue;
                    const float sy = fmaxf(((float)oy + 0.5f) * rs - 0.5f, 0.f);
                    const int y0 = min((int)sy, H - 1);
                    const float ly = sy - (float)y0, hy = 1.f - ly;
                    // y0 is yb + i - 1 for the upper half of the S output rows, yb + i for the lower (clamped rows coincide)
                    const bool low = (y0 - (yb - 1)) > i;
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float top = low ? hrow[i + 1][k] : hrow[i][k];
                        const float bot = low ? hrow[i + 2][k] : hrow[i + 1][k];
                        o[k] = hy * top + ly * bot;
                    }
                    if (O16) {
                        // the fp32 result first, THEN one rounding to fp16 (pinned: hipcc otherwise folds the blend into v_fma_mixlo_f16,
                        // a single rounding from the exact product sum — not the bits of the fp32 kernel's output rounded to fp16)
#pragma unroll
                        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(o[k]));
                        const half4 hv = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                        half4* d16 = reinterpret_cast<half4*>(op16 + (size_t)oy * OW + q * 4);
                        if (NT) __builtin_nontemporal_store(hv, d16);
                        else *d16 = hv;
                        continue;
                    }
                    float* dst = op + (size_t)oy * OW + q * 4;
                    if (NT)
                        __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(dst));
                    else
                        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
                }
            }
            if (sub + 1 < SUBS) {  // slide the window: the last two rows stay, UP_ROWS new ones come in
#pragma unroll
                for (int k = 0; k < 4; ++k) { hrow[0][k] = hrow[UP_ROWS][k]; hrow[1][k] = hrow[UP_ROWS + 1][k]; }
#pragma unroll
                for (int r = 0; r < UP_ROWS; ++r) hinterp(nv[r], hrow[2 + r]);
            }
        }
    }
}

#ifdef VKN_DEBUG  // k_upsample_f (fill-pattern x4 upsample; measured slower): tools/experiments/upsample_fill.inc
#include "../../tools/experiments/upsample_fill.inc"
#endif

int vkn_launch_upsample(const float* in, float* out, int planes, int H, int W, int S, hipStream_t stream, int out_f16) {
    if (S < 1) return VKN_E_SHAPE;
    if (out_f16) {   // fp16 output: the x2 / x4 staged kernels only (what every shipped config uses), 8-byte aligned quads
        if ((S != 2 && S != 4) || ((W * S) % 4) != 0 || (reinterpret_cast<uintptr_t>(out) & 7)) return VKN_E_SHAPE;
        int done16 = 0;
        while (done16 < planes) {
            const int chunk = (planes - done16 > 32768) ? 32768 : planes - done16;
            const float* ip = in + (size_t)done16 * H * W;
            float* op = reinterpret_cast<float*>(reinterpret_cast<_Float16*>(out) + (size_t)done16 * H * S * W * S);
            dim3 grid((H + UP_ROWS * 4 - 1) / (UP_ROWS * 4), chunk);
            if (S == 4) hipLaunchKernelGGL((k_upsample_s<4, 1, 4, 1>), grid, dim3(256), 0, stream, ip, op, H, W);
            else hipLaunchKernelGGL((k_upsample_s<2, 1, 4, 1>), grid, dim3(256), 0, stream, ip, op, H, W);
            VKN_CHECK_LAUNCH();
            done16 += chunk;
        }
        return VKN_OK;
    }
    int done = 0;
    while (done < planes) {  // gridDim.y <= 65535
        const int chunk = (planes - done > 32768) ? 32768 : planes - done;
        const float* ip = in + (size_t)done * H * W;
        float* op = out + (size_t)done * H * S * W * S;
        const int mode = vkn_dbg_env("VKN_UPSAMPLE", 14);  // debug build only: 0 = generic kernel; 1x/2x = staged nt/plain stores, x = 1|4 groups
        const bool staged = mode != 0 && (S == 2 || S == 4) && ((W * S) % 4) == 0 && (reinterpret_cast<uintptr_t>(op) & 15) == 0;
#ifdef VKN_DEBUG
        if (mode >= 100 && S == 4 && (W % 64) == 0 && (reinterpret_cast<uintptr_t>(op) & 15) == 0) {   // k_upsample_f A/B: 1 <R code> <xmap*2 + nt>
            const int rc = (mode / 10) % 10, xm = (mode % 10) >> 1, ntv = mode & 1;
            const int R = rc == 0 ? 2 : rc == 1 ? 4 : 8;
            const int ngroups = (H * 4 + 8 * R - 1) / (8 * R);
            dim3 grid(xm ? ((ngroups + 7) / 8) * 64 : ngroups * 8, chunk);
#define UF_LAUNCH(RV, XV, NV) hipLaunchKernelGGL((k_upsample_f<RV, XV, NV>), grid, dim3(256), 0, stream, ip, op, H, W)
            if (R == 2) { if (xm) { if (ntv) UF_LAUNCH(2, 1, 1); else UF_LAUNCH(2, 1, 0); } else { if (ntv) UF_LAUNCH(2, 0, 1); else UF_LAUNCH(2, 0, 0); } }
            else if (R == 4) { if (xm) { if (ntv) UF_LAUNCH(4, 1, 1); else UF_LAUNCH(4, 1, 0); } else { if (ntv) UF_LAUNCH(4, 0, 1); else UF_LAUNCH(4, 0, 0); } }
            else { if (xm) { if (ntv) UF_LAUNCH(8, 1, 1); else UF_LAUNCH(8, 1, 0); } else { if (ntv) UF_LAUNCH(8, 0, 1); else UF_LAUNCH(8, 0, 0); } }
#undef UF_LAUNCH
        } else
#endif
        if (staged) {
            const bool nt = mode / 10 != 2;
            const int subs = (mode % 10 == 1) ? 1 : 4;
            dim3 grid((H + UP_ROWS * subs - 1) / (UP_ROWS * subs), chunk);
#define UP_LAUNCH(SV, NTV, SUBV) hipLaunchKernelGGL((k_upsample_s<SV, NTV, SUBV>), grid, dim3(256), 0, stream, ip, op, H, W)
#ifdef VKN_DEBUG
            if (S == 4 && mode / 10 == 5) { hipLaunchKernelGGL((k_upsample_s<4, 5, 4>), grid, dim3(256), (size_t)(UP_ROWS * 4 + 2) * W * sizeof(float), stream, ip, op, H, W); }  // input rows through LDS (A/B)
            else if (S == 4 && mode / 10 == 6) { UP_LAUNCH(4, 3, 4); }  // nontemporal input loads (A/B)
            else if (S == 4 && mode / 10 == 7) { UP_LAUNCH(4, 4, 4); }  // all input rows requested up front (A/B)
            else if (S == 4 && mode / 10 == 3) { UP_LAUNCH(4, 2, 4); }  // write-only ablation: the store pattern's own ceiling
            else if (S == 4 && (mode % 10 == 8 || mode % 10 == 9)) {  // 8 / 32 row groups per workgroup (quarter / whole plane)
                if (mode % 10 == 8) { grid.x = (H + UP_ROWS * 8 - 1) / (UP_ROWS * 8); UP_LAUNCH(4, 1, 8); }
                else { grid.x = (H + UP_ROWS * 32 - 1) / (UP_ROWS * 32); UP_LAUNCH(4, 1, 32); }
            } else
#endif
            if (S == 4) {
                if (nt && subs == 4) UP_LAUNCH(4, 1, 4);
                else if (nt) UP_LAUNCH(4, 1, 1);
                else if (subs == 4) UP_LAUNCH(4, 0, 4);
                else UP_LAUNCH(4, 0, 1);
            } else {
                if (nt && subs == 4) UP_LAUNCH(2, 1, 4);
                else if (nt) UP_LAUNCH(2, 1, 1);
                else if (subs == 4) UP_LAUNCH(2, 0, 4);
                else UP_LAUNCH(2, 0, 1);
            }
#undef UP_LAUNCH
        } else {
            // generic scale: non-temporal stores, 4 input rows per workgroup (tools/upsample_ab.py: plain stores 650-690 us,
            // nt 1 row 517 us, nt 4 rows 478 us, nt 16 rows 492 us at cfg2)
            hipLaunchKernelGGL((k_upsample<1, 4>), dim3((H + 3) / 4, chunk), dim3(256), 0, stream, ip, op, H, W, S);
        }
        VKN_CHECK_LAUNCH();
        done += chunk;
    }
    return VKN_OK;
}
