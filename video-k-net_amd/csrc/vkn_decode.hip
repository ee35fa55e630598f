// vkn_decode.hip — mask decode: out[b][n][p] = sum_c K[b][n][c] * x[b][c][p] (+ kb[b][n])
//
// Replaces the reference's per-image `F.conv2d(mask_x[i:i+1], mask_feat[i])` with K=1
// (knet/det/kernel_update_head.py:247-260; video: knet/video/kernel_update_head.py:506-519), with the
// per-stage 1x1 `feat_transform` conv (:107-117,179-180) folded into K and kb by the update kernels.
//
// MI355X design: HBM-bound stream of x (read once) and of the logits (written once).
//   * one persistent 512-thread workgroup per CU; the frame's N decode kernels live in LDS as two f16 planes
//     (hi/lo split of the fp32 values, 2^-22 relative), padded rows -> conflict-free ds_read_b128;
//   * each wave owns 32-pixel strips: B operand = x, loaded straight from global into MFMA fragments
//     (lane = pixel -> 128-B coalesced segments per channel), split to f16 hi/lo in registers;
//     A operand = kernels from LDS; 3 x v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi) per (n-block, 16 ch);
//   * 6-deep register ring of x fragments (5 k-steps = 10 KB in flight per wave) across strip boundaries;
//   * output D[n][px]: lanes 0..31 store 128 contiguous bytes of one mask row.
#include "vkn_common.h"
#include "vkn_launch.h"

#define DEC_THREADS 512
#define DEC_WAVES 8

template <int NB>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decode_mfma(
    const float* __restrict__ x, const _Float16* __restrict__ kfh, const _Float16* __restrict__ kfl,
    const float* __restrict__ kb, float* __restrict__ out, int N, int NPT, int n0, int C, int P, int px_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDK = C + 8;  // halfs per LDS row; (C+8)*2 B = odd multiple of 16 B -> b128 reads conflict-free
    _Float16* ldsH = reinterpret_cast<_Float16*>(smem);
    _Float16* ldsL = ldsH + NB * 32 * LDK;

    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar control flow
    const int g = lane >> 5, li = lane & 31;

    // ---- stage this frame's kernels (rows n0 .. n0+NB*32) into LDS
    {
        const _Float16* gh = kfh + ((size_t)b * NPT + n0) * C;
        const _Float16* gl = kfl + ((size_t)b * NPT + n0) * C;
        const int cpr = C >> 3;
        for (int i = threadIdx.x; i < NB * 32 * cpr; i += DEC_THREADS) {
            const int r = i / cpr, q = i - r * cpr;
            half8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n0 + r < N) {  // rows >= N of the planes are never written by the producer: treat as zero
                vh = *reinterpret_cast<const half8*>(gh + (size_t)r * C + q * 8);
                vl = *reinterpret_cast<const half8*>(gl + (size_t)r * C + q * 8);
            }
            *reinterpret_cast<half8*>(ldsH + r * LDK + q * 8) = vh;
            *reinterpret_cast<half8*>(ldsL + r * LDK + q * 8) = vl;
        }
    }
    // folded decode bias of the chunk's rows -> LDS (accumulators start from it)
    float* kbs = reinterpret_cast<float*>(ldsL + NB * 32 * LDK);
    if (threadIdx.x < NB * 32) {
        const int n = n0 + threadIdx.x;
        kbs[threadIdx.x] = (kb && n < N) ? kb[(size_t)b * N + n] : 0.f;
    }
    __syncthreads();

    const int p_begin = blockIdx.x * px_per_wg;
    const int p_end = min(P, p_begin + px_per_wg);
    const int nstrips = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;
    const int my = (nstrips > wave) ? (nstrips - wave + DEC_WAVES - 1) / DEC_WAVES : 0;
    const int KS = C >> 4;
    const int total = my * KS;
    if (total == 0) return;

    const float* xb = x + (size_t)b * C * P;
    float* ob = out + (size_t)b * N * P;

    f32x16 acc[NB];
    float r0[8], r1[8], r2[8], r3[8], r4[8], r5[8];
    int ld_ks = 0, ld_sl = 0, ld_cnt = 0;  // next fragment to load
    int c_ks = 0, c_sl = 0;                // next fragment to consume

#define DEC_LOAD(REG)                                                                       \
    do { /* unconditional: past the end it re-reads the last fragment (keeps vmcnt counting exact) */ \
        int px_ = p_begin + ((wave + DEC_WAVES * ld_sl) << 5) + li;                         \
        px_ = min(px_, P - 1);                                                              \
        const float* p_ = xb + (size_t)((ld_ks << 4) + (g << 3)) * P + px_;                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) REG[e] = p_[(size_t)e * P];           \
        const bool adv_ = (ld_cnt + 1 < total);                                             \
        const bool wrap_ = (ld_ks + 1 == KS);                                               \
        ld_cnt += adv_ ? 1 : 0;                                                             \
        ld_sl += (adv_ && wrap_) ? 1 : 0;                                                   \
        ld_ks = adv_ ? (wrap_ ? 0 : ld_ks + 1) : ld_ks;                                     \
    } while (0)

#define DEC_COMPUTE(REG)                                                                                  \
    do {                                                                                                  \
        if (c_ks == 0) {                                                                                  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                             \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[nb][r] = kbs[nb * 32 + vkn_cd_row(r, lane)]; \
        }                                                                                                 \
        half8 bh, bl;                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                   \
            _Float16 h_, l_;                                                                              \
            vkn_split_f16(REG[e], h_, l_);                                                                \
            bh[e] = h_;                                                                                   \
            bl[e] = l_;                                                                                   \
        }                                                                                                 \
        const int cb_ = (c_ks << 4) + (g << 3);                                                           \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                               \
            const _Float16* ap_ = ldsH + (nb * 32 + li) * LDK + cb_;                                      \
            const half8 ah = *reinterpret_cast<const half8*>(ap_);                                        \
            const half8 al = *reinterpret_cast<const half8*>(ap_ + NB * 32 * LDK);                        \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nb], 0, 0, 0);                   \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nb], 0, 0, 0);                   \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nb], 0, 0, 0);                   \
        }                                                                                                 \
        if (++c_ks == KS) {                                                                               \
            const int px_ = p_begin + ((wave + DEC_WAVES * c_sl) << 5) + li;                              \
            if (px_ < p_end) {                                                                            \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                       \
                    float* o_ = ob + (size_t)(n0 + nb * 32 + 4 * g) * P + px_;                            \
                    if (n0 + nb * 32 + 32 <= N) { /* full block: no per-row guard (uniform branch) */     \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                    \
                            o_[(size_t)((r & 3) + 8 * (r >> 2)) * P] = acc[nb][r];                        \
                    } else {                                                                              \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                  \
                            const int n_ = n0 + nb * 32 + vkn_cd_row(r, lane);                            \
                            if (n_ < N) o_[(size_t)((r & 3) + 8 * (r >> 2)) * P] = acc[nb][r];            \
                        }                                                                                 \
                    }                                                                                     \
                }                                                                                         \
            }                                                                                             \
            c_ks = 0;                                                                                     \
            ++c_sl;                                                                                       \
        }                                                                                                 \
    } while (0)

    // 6-deep register ring: 5 fragments (40 dword loads = 10 KB per wave, 80 KB per CU) in flight behind the MFMAs
    DEC_LOAD(r0);
    DEC_LOAD(r1);
    DEC_LOAD(r2);
    DEC_LOAD(r3);
    DEC_LOAD(r4);
    for (int f = 0; f < total; f += 6) {
        DEC_LOAD(r5);
        DEC_COMPUTE(r0);
        if (f + 1 >= total) break;
        DEC_LOAD(r0);
        DEC_COMPUTE(r1);
        if (f + 2 >= total) break;
        DEC_LOAD(r1);
        DEC_COMPUTE(r2);
        if (f + 3 >= total) break;
        DEC_LOAD(r2);
        DEC_COMPUTE(r3);
        if (f + 4 >= total) break;
        DEC_LOAD(r3);
        DEC_COMPUTE(r4);
        if (f + 5 >= total) break;
        DEC_LOAD(r4);
        DEC_COMPUTE(r5);
    }
#undef DEC_LOAD
#undef DEC_COMPUTE
}

// Exact-fp32 debug / fallback kernel: one thread per (n, px), k-ordered fmaf chain.
__global__ __launch_bounds__(256) void k_decode_ref(const float* __restrict__ x, const float* __restrict__ kern,
                                                    const float* __restrict__ kb, float* __restrict__ out, int N,
                                                    int C, int P) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y, b = blockIdx.z;
    if (px >= P) return;
    const float* xp = x + (size_t)b * C * P + px;
    const float* kp = kern + ((size_t)b * N + n) * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(kp[c], xp[(size_t)c * P], acc);
    out[((size_t)b * N + n) * P + px] = acc + (kb ? kb[(size_t)b * N + n] : 0.f);
}

// fp32 kernels [B][N][C] -> two f16 planes [B][NPT][C] (rows >= N zero).  Used by the stand-alone decode entry
// point; inside a stage the update kernels write the planes directly.
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ kern, _Float16* __restrict__ kfh,
                                                      _Float16* __restrict__ kfl, int N, int NPT, int C) {
    const int row = blockIdx.x;  // b*NPT + n
    const int b = row / NPT, n = row - b * NPT;
    for (int c = threadIdx.x; c < C; c += 256) {
        _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
        if (n < N) vkn_split_f16(kern[((size_t)b * N + n) * C + c], h, l);
        kfh[(size_t)row * C + c] = h;
        kfl[(size_t)row * C + c] = l;
    }
}

static int dec_set_lds(const void* fn, size_t bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}

// host launcher.  kfh/kfl: [B][NPT][C] f16, NPT = roundup(N,32).  Returns VKN_* code.
int vkn_launch_decode(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B,
                      int N, int C, int P, hipStream_t stream) {
    if (B <= 0 || N <= 0 || P <= 0) return VKN_E_ARG;
    if (C % 16 != 0 || C > 512) return VKN_E_SHAPE;
    const int NPT = (N + 31) / 32 * 32;
    // persistent grid: ~1 workgroup per CU over the whole batch, >= 256 px (8 strips) per workgroup
    int wg_per_frame = 256 / B;
    if (wg_per_frame < 1) wg_per_frame = 1;
    int px_per_wg = (P + wg_per_frame - 1) / wg_per_frame;
    px_per_wg = (px_per_wg + 255) / 256 * 256;
    const int G = (P + px_per_wg - 1) / px_per_wg;
    for (int n0 = 0; n0 < NPT; n0 += 128) {
        const int nb = (NPT - n0 >= 128) ? 4 : (NPT - n0) / 32;
        const size_t lds = (size_t)2 * nb * 32 * (C + 8) * sizeof(_Float16) + (size_t)nb * 32 * sizeof(float);
        dim3 grid(G, B, 1), block(DEC_THREADS);
#define DEC_CASE(NBV)                                                                                          \
    case NBV:                                                                                                  \
        if (dec_set_lds((const void*)k_decode_mfma<NBV>, lds)) return VKN_E_LAUNCH;                            \
        hipLaunchKernelGGL(k_decode_mfma<NBV>, grid, block, lds, stream, x, kfh, kfl, kb, out, N, NPT, n0, C, P, \
                           px_per_wg);                                                                         \
        break;
        switch (nb) {
            DEC_CASE(1)
            DEC_CASE(2)
            DEC_CASE(3)
            DEC_CASE(4)
            default:
                return VKN_E_SHAPE;
        }
#undef DEC_CASE
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}

int vkn_launch_decode_ref(const float* x, const float* kern, const float* kb, float* out, int B, int N, int C, int P,
                          hipStream_t stream) {
    dim3 grid((P + 255) / 256, N, B);
    hipLaunchKernelGGL(k_decode_ref, grid, dim3(256), 0, stream, x, kern, kb, out, N, C, P);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_split_planes(const float* kern, _Float16* kfh, _Float16* kfl, int B, int N, int C, hipStream_t stream) {
    const int NPT = (N + 31) / 32 * 32;
    hipLaunchKernelGGL(k_split_planes, dim3(B * NPT), dim3(256), 0, stream, kern, kfh, kfl, N, NPT, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
