// vkn_decode.hip — mask decode: out[b][n][p] = sum_c K[b][n][c] * x[b][c][p] (+ kb[b][n])
//
// Replaces the reference's per-image `F.conv2d(mask_x[i:i+1], mask_feat[i])` with K=1
// (knet/det/kernel_update_head.py:247-260; video: knet/video/kernel_update_head.py:506-519), with the
// per-stage 1x1 `feat_transform` conv (:107-117,179-180) folded into K and kb by the update kernels.
//
// MI355X design: HBM-bound stream of x (read once) and of the logits (written once).
//   * one persistent 512-thread workgroup per CU; the frame's N decode kernels live in LDS as two f16 planes
//     (hi/lo split of the fp32 values, 2^-22 relative), padded rows -> conflict-free ds_read_b128;
//   * each wave owns 64-pixel tiles = two interleaved 32-column MFMA strips (even / odd pixels): B operand = x, loaded
//     straight from global into MFMA fragments with 8-byte buffer loads (lane = pixel pair -> 256-B coalesced segments per
//     channel row, descriptor + SGPR row offsets, no VALU address math), split to f16 hi/lo in registers;
//     A operand = kernels from LDS, each fragment reused for both strips; 3 x v_mfma_f32_32x32x16_f16
//     (hi*hi + hi*lo + lo*hi) per (strip, n-block, 16 channels), the two strips' accumulators alternating;
//   * 3-deep register ring of x fragments (2 k-steps = 8 KB in flight per wave) across tile boundaries;
//   * output D[n][px]: 8-byte stores, lanes 0..31 write 256 contiguous bytes of one mask row.
#include "vkn_common.h"
#include "vkn_launch.h"

#define DEC_THREADS 512
#define DEC_WAVES 8
#define DEC_TILE 64  // pixels per wave tile: two interleaved 32-column MFMA strips (even / odd pixels)
#define DEC_OPT_DEFAULT 1  // round-2 variants of k_decode_mfma compiled into the release library (see OPT below)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));

// ---- bit-packed output (intermediate stages of the fused head): instead of the logits, emit bit(z >= thr) — all the next
// stage's gather consumes — as words[B][P/64][2][NPT]: for 64-px tile T and row n, word [T][0][n] bit i = pixel 64 T + 2 i
// (even pixels = MFMA strip 0) and word [T][1][n] bit i = pixel 64 T + 2 i + 1 (strip 1).  That is exactly what the two ballots of
// a C/D register deliver, so the epilogue is 2 v_cmp + 4 selects per register pair and four 256-B row stores per tile (a first
// version that interleaved the two ballots into pixel order on the scalar unit cost as much as the logits stores it replaced:
// one SALU serves the whole CU).  The gather's byte -> fragment table absorbs the even/odd split.
// Cuts the stage hand-off from 2 x 15.3 MB to 2 x 0.5 MB per frame.
template <int NB>
__device__ __forceinline__ void dec_emit_bits(const f32x16 (&acc)[2][NB], float thr, unsigned* __restrict__ wbase, int NPT,
                                              int lane) {
    constexpr int NH = (NB * 32 + 63) / 64;
    int w[2][NH];  // [even | odd pixels][row half]: lane = row & 63
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < NH; ++h) w[t][h] = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned long long m0 = __ballot(acc[0][nb][r] >= thr);  // even pixels: bit 32 g + li
            const unsigned long long m1 = __ballot(acc[1][nb][r] >= thr);  // odd pixels
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int row = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;  // compile-time
                const bool mine = lane == (row & 63);                      // lane <- its row's words (values are wave-uniform)
                w[0][row >> 6] = mine ? (int)(unsigned)(m0 >> (32 * g)) : w[0][row >> 6];
                w[1][row >> 6] = mine ? (int)(unsigned)(m1 >> (32 * g)) : w[1][row >> 6];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep each pair of ballots next to its selects (else 128 masks spill SGPRs)
        }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < NH; ++h)
            if (h * 64 + lane < NB * 32) wbase[(size_t)t * NPT + h * 64 + lane] = (unsigned)w[t][h];
}

// ABL (ablation, debugging only; selected by env VKN_DECODE_ABL): 0 = the real kernel, 1 = no MFMA, 2 = no x loads,
// 3 = no output stores.  Variants 1-3 produce WRONG results by construction and exist to attribute time.
// OPT (bit mask, round-2 variants measured with tools/perf_r02.py): 1 = request the first x fragments BEFORE staging the kernel
// planes (the prologue overlaps their latency); 2 = static `s_setprio 1` for the younger half of the workgroup (waves 4-7);
// 4 = 16-byte stores: adjacent lanes exchange half of their pixel pairs (DPP quad_perm) so that a lane stores 4 consecutive
// pixels of ONE row — half the store instructions, same 256-byte row segments.
// XH (x storage): 0 = fp32 (split to f16 hi / lo in registers, 3 MFMAs per operand pair); 1 = fp16, 2 = bf16 (converted to f16:
// exact for 2^-14 <= |x| < 65504): x IS its own high half, the low half is zero — a lane loads its pixel pair as ONE dword, and the
// two MFMAs against x_lo disappear.  The remaining sequence (K_hi x, K_lo x per pixel) is the fp32 kernel's with the zero terms
// removed, so on x' = float(half(x)) both kernels return the same bits.
// OSC = 1 (the backward passes of the training step: gradients are scaled by a power of two into the f16 range and back): every output is
// multiplied by the device scalar *oscale before it is stored — instead of a second pass over the result.  The inference
// instantiations (OSC = 0) are the kernels they were.
template <int NB, int ABL, int RING, int BITS, int OPT, int XH = 0, int OSC = 0>
__global__ __launch_bounds__(DEC_THREADS, 2) void k_decode_mfma(
    const float* __restrict__ x, const _Float16* __restrict__ kfh, const _Float16* __restrict__ kfl,
    const float* __restrict__ kb, float* __restrict__ out, int N, int NPT, int n0, int C, int P, int px_per_wg,
    int xcd_remap, VknDecodeStrides fs, unsigned* __restrict__ bits_out, float thr, const float* __restrict__ oscale) {
    const float osc_ = OSC ? *oscale : 1.f;
#define DEC_V(v) (OSC ? (v) * osc_ : (v))
    n0 += (int)blockIdx.z * NB * 32;   // few frames per launch: the kernel rows are spread over gridDim.z workgroups per pixel range (round 5)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDK = C + 8;  // halfs per LDS row; (C+8)*2 B = odd multiple of 16 B -> b128 reads conflict-free
    _Float16* ldsH = reinterpret_cast<_Float16*>(smem);
    _Float16* ldsL = ldsH + NB * 32 * LDK;

    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar control flow
    const int g = lane >> 5, li = lane & 31;

    float* kbs = reinterpret_cast<float*>(ldsL + NB * 32 * LDK);
    auto stage_planes = [&]() {
        // ---- stage this frame's kernels (rows n0 .. n0+NB*32) into LDS
        {
            const _Float16* gh = kfh + (size_t)b * fs.plane + (size_t)n0 * C;
            const _Float16* gl = kfl + (size_t)b * fs.plane + (size_t)n0 * C;
            const int cpr = C >> 3;
            // FOUR iterations' loads (8 x 16 bytes per thread) are requested before the first LDS store: the rolled load -> store loop
            // was one memory round trip per iteration — eight in a row at C = 256, most of this kernel's time at one frame per launch
            const int items = NB * 32 * cpr;
            for (int i0 = threadIdx.x; i0 < items; i0 += 4 * DEC_THREADS) {
                half8 vh[4], vl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * DEC_THREADS;
                    const int r = i / cpr, q = i - r * cpr;
                    vh[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    vl[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (i < items && n0 + r < N) {  // rows >= N of the planes are never written by the producer: treat as zero
                        vh[u] = *reinterpret_cast<const half8*>(gh + (size_t)r * C + q * 8);
                        vl[u] = *reinterpret_cast<const half8*>(gl + (size_t)r * C + q * 8);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * DEC_THREADS;
                    if (i < items) {
                        const int r = i / cpr, q = i - r * cpr;
                        *reinterpret_cast<half8*>(ldsH + r * LDK + q * 8) = vh[u];
                        *reinterpret_cast<half8*>(ldsL + r * LDK + q * 8) = vl[u];
                    }
                }
            }
        }
        // folded decode bias of the chunk's rows -> LDS (accumulators start from it)
        if (threadIdx.x < NB * 32) {
            const int n = n0 + threadIdx.x;
            kbs[threadIdx.x] = (kb && n < N) ? kb[(size_t)b * fs.kb + n] : 0.f;
        }
        __syncthreads();
    };
    if (!(OPT & 1)) stage_planes();

    // XCD-aware pixel ranges (workgroup id % 8 = XCD, observed): `xcd_remap` gives each XCD a contiguous 1/8 of the frame
    int gx = blockIdx.x;
    if (xcd_remap && (gridDim.x % 8) == 0) gx = (gx % 8) * (gridDim.x / 8) + gx / 8;
    const int p_begin = gx * px_per_wg;
    const int p_end = min(P, p_begin + px_per_wg);
    const int ntile = (p_end > p_begin) ? (p_end - p_begin + DEC_TILE - 1) / DEC_TILE : 0;
    const int my = (ntile > wave) ? (ntile - wave + DEC_WAVES - 1) / DEC_WAVES : 0;
    const int KS = C >> 4;
    const int total = my * KS;
    if (total == 0) {
        if (OPT & 1) stage_planes();  // every wave takes part in the staging barrier
        return;
    }
    if ((OPT & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);  // `wave` is provably uniform: a scalar branch around one s_setprio

    // Buffer descriptors built from uniform values only: loads/stores are `buffer_* v, v_off, s[rsrc], s_off offen` with ONE
    // per-lane VGPR offset and every row / tile offset in the SGPR operand (no per-load VALU address arithmetic).
    // NOTE: hipcc (ROCm 7.2) miscompiles `__builtin_bit_cast(T, vec[i])` on an ext_vector ELEMENT (always yields element 0,
    // tools/scratch/buftest.hip) — elements are copied to scalars first and converted with __uint_as_float / __float_as_uint.
    const __amdgpu_buffer_rsrc_t xrs =
        XH ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P), 0,
                                               C * P * 2, 0x00020000)
           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * C * P), 0, C * P * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * fs.out, 0, N * P * 4, 0x00020000);

    f32x16 acc[2][NB];
    u32x2 r0[8], r1[8], r2[8];
    int ld_ks = 0, ld_sl = 0, ld_cnt = 0;  // next fragment to load
    int c_ks = 0, c_sl = 0;                // next fragment to consume

    // lane (g, li) of a tile at pixel p0: pixels p0 + 2*li (+1), channels ks*16 + 8*g + e.  Near the frame end the pixel pair
    // is clamped into the row (such lanes are never stored).
#define DEC_LOAD(REG)                                                                                            \
    do { /* unconditional: past the end it re-reads the last fragment (keeps vmcnt counting exact) */            \
        const int p0_ = p_begin + (wave + DEC_WAVES * ld_sl) * DEC_TILE;                                         \
        const int voff_ = (((g << 3) * P + min(2 * li, max(P - 2 - p0_, 0))) << (XH ? 1 : 2));                   \
        const int soff_ = ((ld_ks << 4) * P + p0_) << (XH ? 1 : 2);                                              \
        if (XH && (OPT & 8)) { /* half storage, paired lanes (debug A/B): lane li loads 4 px x 4 channels (8 B), see DEC_COMPUTE */ \
            const int vp_ = ((((g << 3) + ((li & 1) << 2)) * P + min(4 * (li >> 1), max(P - 4 - p0_, 0))) << 1);  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
                REG[e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, vp_, soff_ + ((e * P) << 1), 3);              \
        } else if (XH) { /* half storage: the pixel pair is one dword */                                         \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                        \
                REG[e] = u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff_, soff_ + ((e * P) << 1), 3), 0u}; \
        } else if (VKN_ABL_IS(ABL, 2)) {                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) REG[e] = u32x2{(unsigned)(voff_ + e), (unsigned)soff_}; \
        } else {                                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                        \
                /* aux = 2 (nt): x is streamed once per launch; measured +9 % (92 -> 84 us, cfg2 B = 8).  ABL 4 = plain */ \
                /* cache policy aux = 3 (sc0 | nt): x is streamed once per launch.  tools/decode_sweep.py, cfg2 B = 8:        */ \
                /* plain 88.7-94.6 us, nt 81.9-86.2 us, sc0|nt 78.3-82.7 us.  ABL 4 = plain, ABL 6 = nt only (A/B)     */ \
                REG[e] = VKN_ABL_IS(ABL, 4) ? __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 0) \
                         : VKN_ABL_IS(ABL, 6) ? __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 2) \
                         : VKN_ABL_IS(ABL, 7) ? __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 16) /* sc1 */ \
                         : VKN_ABL_IS(ABL, 8) ? __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 18) /* sc1 nt */ \
                         : VKN_ABL_IS(ABL, 9) ? __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 19) /* sc0 sc1 nt */ \
                                      : __builtin_amdgcn_raw_buffer_load_b64(xrs, voff_, soff_ + ((e * P) << 2), 3); \
        }                                                                                                        \
        const bool adv_ = (ld_cnt + 1 < total);                                                                  \
        const bool wrap_ = (ld_ks + 1 == KS);                                                                    \
        ld_cnt += adv_ ? 1 : 0;                                                                                  \
        ld_sl += (adv_ && wrap_) ? 1 : 0;                                                                        \
        ld_ks = adv_ ? (wrap_ ? 0 : ld_ks + 1) : ld_ks;                                                          \
    } while (0)

#define DEC_COMPUTE(REG)                                                                                          \
    do {                                                                                                          \
        if (c_ks == 0) {                                                                                          \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                     \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                  \
                    const float kb_ = kbs[nb * 32 + vkn_cd_row(r, lane)];                                         \
                    acc[0][nb][r] = kb_;                                                                          \
                    acc[1][nb][r] = kb_;                                                                          \
                }                                                                                                 \
        }                                                                                                         \
        half8 bh0, bl0, bh1, bl1;                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                           \
            _Float16 h_, l_;                                                                                      \
            unsigned u0_ = REG[e][0], u1_ = REG[e][1];                                                            \
            if (XH && (OPT & 8)) { /* lanes 2j / 2j+1 hold channels 0-3 / 4-7 of pixels 4j..4j+3: swap halves via DPP */ \
                const int e4_ = e & 3, odd_ = li & 1;                                                             \
                const unsigned own0_ = REG[e4_][0], own1_ = REG[e4_][1];                                          \
                const unsigned recv_ = (unsigned)__builtin_amdgcn_mov_dpp((int)(odd_ ? own0_ : own1_), 0xB1, 0xF, 0xF, true); \
                u0_ = (e < 4) ? (odd_ ? recv_ : own0_) : (odd_ ? own1_ : recv_);                                  \
            }                                                                                                     \
            if (XH == 1) { /* fp16 pair: low half = even pixel */                                                 \
                bh0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0_ & 0xFFFFu));                           \
                bh1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0_ >> 16));                               \
            } else if (XH == 2) { /* bf16 pair -> fp32 (exact) -> f16 */                                          \
                bh0[e] = (_Float16)__uint_as_float(u0_ << 16);                                                    \
                bh1[e] = (_Float16)__uint_as_float(u0_ & 0xFFFF0000u);                                            \
            } else {                                                                                              \
                vkn_split_f16(__uint_as_float(u0_), h_, l_);                                                      \
                bh0[e] = h_;                                                                                      \
                bl0[e] = l_;                                                                                      \
                vkn_split_f16(__uint_as_float(u1_), h_, l_);                                                      \
                bh1[e] = h_;                                                                                      \
                bl1[e] = l_;                                                                                      \
            }                                                                                                     \
        }                                                                                                         \
        const int cb_ = (c_ks << 4) + (g << 3);                                                                   \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                       \
            const _Float16* ap_ = ldsH + (nb * 32 + li) * LDK + cb_;                                              \
            const half8 ah = *reinterpret_cast<const half8*>(ap_);                                                \
            const half8 al = *reinterpret_cast<const half8*>(ap_ + NB * 32 * LDK);                                \
            if (VKN_ABL_IS(ABL, 1)) {                                                                                   \
                asm volatile("" ::"v"(ah), "v"(al), "v"(bh0), "v"(bl0), "v"(bh1), "v"(bl1));                      \
            } else { /* alternate the two accumulators so dependent MFMAs are never back to back */               \
                acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc[0][nb], 0, 0, 0);                \
                acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, acc[1][nb], 0, 0, 0);                \
                if (!XH) {                                                                                        \
                    acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc[0][nb], 0, 0, 0);            \
                    acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, acc[1][nb], 0, 0, 0);            \
                }                                                                                                 \
                acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc[0][nb], 0, 0, 0);                \
                acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, acc[1][nb], 0, 0, 0);                \
            }                                                                                                     \
        }                                                                                                         \
        if (++c_ks == KS) {                                                                                       \
            const int p0_ = p_begin + (wave + DEC_WAVES * c_sl) * DEC_TILE;                                       \
            const int px_ = p0_ + 2 * li;                                                                         \
            const int vst_ = ((4 * g) * P + 2 * li) << 2;                                                         \
            /* the 64 row offsets row * P * 4 are loop invariants the compiler would keep in 64 SGPRs (-> 230 SGPR spills as   */ \
            /* v_writelane / v_readlane pairs in this loop); an opaque copy of P makes it recompute them per store: one s_mul   */ \
            int Pq_ = P;                                                                                          \
            asm volatile("" : "+s"(Pq_));                                                                         \
            if (BITS) { /* P % 64 == 0 (launcher): every tile is whole */                                        \
                dec_emit_bits<NB>(acc, thr, bits_out + ((size_t)b * (P >> 5) + ((p0_ >> 6) << 1)) * NPT + n0, NPT, lane);   \
            } else if (!VKN_ABL_IS(ABL, 3) || acc[0][0][0] == 12345.678f) {                                                  \
                if (p0_ + DEC_TILE <= p_end) { /* whole tile in range (uniform): 8-byte stores, 256 B per row */  \
                    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                           \
                        if ((OPT & 4) && n0 + nb * 32 + 32 <= N) { /* 16-byte stores: lanes 2j / 2j+1 swap halves */ \
                            const int q_ = li & 1;                                                                \
                            const int vw_ = ((4 * g + q_) * P + 2 * (li - q_)) << 2;                              \
                            _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                   \
                                const int row_ = n0 + nb * 32 + (r & 3) + 8 * (r >> 2); /* even lanes: row_, odd: row_ + 1 */ \
                                const float e0_ = DEC_V(acc[0][nb][r]), e1_ = DEC_V(acc[1][nb][r]); /* row_: this lane's px pair */ \
                                const float o0_ = DEC_V(acc[0][nb][r + 1]), o1_ = DEC_V(acc[1][nb][r + 1]); /* row_ + 1 */ \
                                /* send what the partner stores: the even lane its row_+1 pair, the odd lane its row_ pair */ \
                                const int s0_ = __float_as_int(q_ ? e0_ : o0_), s1_ = __float_as_int(q_ ? e1_ : o1_); \
                                const int t0_ = __builtin_amdgcn_mov_dpp(s0_, 0xB1, 0xF, 0xF, true); /* quad_perm [1,0,3,2] */ \
                                const int t1_ = __builtin_amdgcn_mov_dpp(s1_, 0xB1, 0xF, 0xF, true);              \
                                u32x4w v_;                                                                        \
                                v_[0] = q_ ? (unsigned)t0_ : __float_as_uint(e0_);                                \
                                v_[1] = q_ ? (unsigned)t1_ : __float_as_uint(e1_);                                \
                                v_[2] = q_ ? __float_as_uint(o0_) : (unsigned)t0_;                                \
                                v_[3] = q_ ? __float_as_uint(o1_) : (unsigned)t1_;                                \
                                __builtin_amdgcn_raw_buffer_store_b128(v_, ors, vw_, (row_ * Pq_ + p0_) << 2, 0);   \
                            }                                                                                     \
                        } else { /* no per-row guard: `ors` covers exactly the frame's N rows, so the stores of rows >= N */ \
                                 /* (ragged last n-block) are dropped by the buffer's range check — no exec branches in the loop */ \
                            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
                                const int row_ = n0 + nb * 32 + (r & 3) + 8 * (r >> 2);                           \
                                const float a0_ = DEC_V(acc[0][nb][r]), a1_ = DEC_V(acc[1][nb][r]);               \
                                const u32x2 v_ = {__float_as_uint(a0_), __float_as_uint(a1_)};                    \
                                if (VKN_ABL_IS(ABL, 5))                                                              \
                                    __builtin_amdgcn_raw_buffer_store_b64(v_, ors, vst_, (row_ * Pq_ + p0_) << 2, 2); \
                                else                                                                              \
                                    __builtin_amdgcn_raw_buffer_store_b64(v_, ors, vst_, (row_ * Pq_ + p0_) << 2, 0); \
                            }                                                                                     \
                        }                                                                                         \
                    }                                                                                             \
                } else { /* ragged tile at the end of the range: per-pixel guards, 4-byte buffer stores */        \
                    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                           \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                          \
                            const int row_ = n0 + nb * 32 + (r & 3) + 8 * (r >> 2);                               \
                            const int so_ = (row_ * Pq_ + p0_) << 2;                                                \
                            if (row_ + 4 * g < N) {                                                               \
                                const float a0_ = DEC_V(acc[0][nb][r]), a1_ = DEC_V(acc[1][nb][r]);               \
                                if (px_ < p_end)                                                                  \
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a0_), ors, vst_, so_, 0); \
                                if (px_ + 1 < p_end)                                                              \
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a1_), ors, vst_ + 4, so_, 0); \
                            }                                                                                     \
                        }                                                                                         \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
            c_ks = 0;                                                                                             \
            ++c_sl;                                                                                               \
        }                                                                                                         \
    } while (0)

    // register ring: RING - 1 fragments (8 dwordx2 loads = 4 KB per wave each) in flight behind the MFMAs
    if constexpr (RING == 3) {
        DEC_LOAD(r0);
        DEC_LOAD(r1);
        if (OPT & 1) stage_planes();
        for (int f = 0; f < total; f += 3) {
            DEC_LOAD(r2);
            DEC_COMPUTE(r0);
            if (f + 1 >= total) break;
            DEC_LOAD(r0);
            DEC_COMPUTE(r1);
            if (f + 2 >= total) break;
            DEC_LOAD(r1);
            DEC_COMPUTE(r2);
        }
    } else {
        u32x2 r3[8];
        DEC_LOAD(r0);
        DEC_LOAD(r1);
        DEC_LOAD(r2);
        if (OPT & 1) stage_planes();
        for (int f = 0; f < total; f += 4) {
            DEC_LOAD(r3);
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
        }
    }
#undef DEC_LOAD
#undef DEC_COMPUTE
#undef DEC_V
}

// NOTE (negative result, round 1): a software-pipelined variant that parked a finished strip's accumulators in a second
// register set and issued its stores a few per k-step during the next strip (so loads, MFMAs and stores interleave inside
// every wave) was built and validated, and measured SLOWER (113 us vs 94 us at cfg2, B = 8): the read and write streams
// already share the memory system at ~4.2 TB/s combined whatever their interleaving (tools/decode_ablation.py).

#ifdef VKN_DEBUG  // rejected / time-attribution variants live outside the product sources
#include "../../tools/experiments/decode_variants.inc"
#endif

// Exact-fp32 debug / fallback kernel: one thread per (n, px), k-ordered fmaf chain.
__global__ __launch_bounds__(256) void k_decode_ref(const float* __restrict__ x, const float* __restrict__ kern,
                                                    const float* __restrict__ kb, float* __restrict__ out, int N,
                                                    int C, int P, VknDecodeStrides fs) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y, b = blockIdx.z;
    if (px >= P) return;
    const float* xp = x + (size_t)b * C * P + px;
    const float* kp = kern + (size_t)b * fs.plane + (size_t)n * C;  // fs.plane: fp32 kernel elements per frame here
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(kp[c], xp[(size_t)c * P], acc);
    out[(size_t)b * fs.out + (size_t)n * P + px] = acc + (kb ? kb[(size_t)b * fs.kb + n] : 0.f);
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

// host launcher.  kfh/kfl: [B][NPT][C] f16, NPT = roundup(N,32).  Returns VKN_* code.
int vkn_launch_decode(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B,
                      int N, int C, int P, hipStream_t stream, int xdt) {
    return vkn_launch_decode_ex(x, kfh, kfl, kb, out, B, N, C, P, 0, N, stream, xdt);
}

// shared != 0: ONE set of kernels / bias for every frame (planes [NPT][C], kb [N]); out_rows: rows per frame of the output
// tensor the N decoded rows are written into (>= N: the caller points `out` at the first of its rows).
static int decode_launch(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B, int N,
                         int C, int P, int shared, int out_rows, unsigned* bits_out, float thr, hipStream_t stream, int xdt,
                         const float* oscale = nullptr);

// xdt: storage type of x (VKN_X_F32 / VKN_X_F16 / VKN_X_BF16); for the half types `x` points at 2-byte elements
int vkn_launch_decode_ex(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B,
                         int N, int C, int P, int shared, int out_rows, hipStream_t stream, int xdt, const float* oscale) {
    return decode_launch(x, kfh, kfl, kb, out, B, N, C, P, shared, out_rows, nullptr, 0.f, stream, xdt, oscale);
}

// bit-packed variant: words [B][P/64][2][roundup(N,32)] (even / odd pixels of each 64-px tile) of bit(logit >= thr)
// instead of the logits (P % 64 == 0)
int vkn_launch_decode_bits(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, unsigned* bits_out,
                           float thr, int B, int N, int C, int P, hipStream_t stream, int xdt) {
    if (!bits_out || (P % 64) != 0) return VKN_E_SHAPE;
    return decode_launch(x, kfh, kfl, kb, nullptr, B, N, C, P, 0, N, bits_out, thr, stream, xdt);
}

static int decode_launch(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float* out, int B, int N,
                         int C, int P, int shared, int out_rows, unsigned* bits_out, float thr, hipStream_t stream, int xdt,
                         const float* oscale) {
    if (oscale && bits_out) return VKN_E_ARG;
    if (B <= 0 || N <= 0 || P <= 0 || out_rows < N) return VKN_E_ARG;
    if (xdt < 0 || xdt > 2) return VKN_E_ARG;
    if (C % 16 != 0 || C > 512 || P < 2 || (P & 1)) return VKN_E_SHAPE;  // odd P: rows not 8-byte aligned (use the ref kernel)
    if ((size_t)C * P * 4 >= ((size_t)1 << 31) || (size_t)N * P * 4 >= ((size_t)1 << 31)) return VKN_E_SHAPE;  // 32-bit buffer offsets
    const int NPT = (N + 31) / 32 * 32;
    const VknDecodeStrides fs{shared ? 0 : (long long)NPT * C, shared ? 0 : (long long)N, (long long)out_rows * P};
    // ONE workgroup per CU over the whole batch whenever the batch is large enough (round 2, tools/perf_r02.py, cfg2, early x loads):
    // a persistent workgroup stages the kernel planes once and its fragment ring runs on across tiles, while every workgroup
    // boundary costs a store drain + plane staging + first-load latency.  Measured px / workgroup -> us: B = 32: 512 -> 329,
    // 2048 -> 332, 4096 (256 WGs) -> 309; B = 16: 512 -> 161, 2048 (256 WGs) -> 153; B = 8: 512 -> 87, 1024 (256 WGs) -> 88;
    // B <= 4: 512 is best (fewer than 256 WGs either way).  Intermediate sizes (2-4 WGs per CU in sequence) are the slowest.
    long long ppx = ((long long)B * P + 255) / 256;
    int px_per_wg = (int)((ppx + 511) / 512 * 512);  // 8 waves x 64-px tiles
    if (px_per_wg < 512) px_per_wg = 512;
    const int ppw_dbg = vkn_dbg_env("VKN_DECODE_PXWG", 0);  // debug build only: override pixels per workgroup
    if (ppw_dbg >= 512) px_per_wg = ppw_dbg / 512 * 512;
    const int G2 = (P + px_per_wg - 1) / px_per_wg;
    const int xcd = vkn_dbg_env("VKN_DECODE_XCD", 1);  // measured +1 % (tools/decode_sweep.py)
    // (debug build) 16-byte variant k_decode4: whole 128-px tiles and 16-byte aligned rows; the logits output only
#ifdef VKN_DEBUG
    const bool wide = xdt == 0 && !bits_out && (P % D4_TILE) == 0 && (C % 64) == 0 && vkn_dbg_env("VKN_DECODE4", 0) != 0 &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
#endif
    // One or two frames per launch leave most CUs without a workgroup (64 workgroups per frame of 128x256) and every workgroup stages ALL
    // kernel rows (120 KB of planes — at one frame as many bytes as the features).  There the rows are spread over blockIdx.z instead:
    // 1 or 2 n-blocks per workgroup, 4 or 2 workgroups per pixel range (x is re-read from L2 / the memory-side cache by each); the
    // accumulators of an n-block see the same MFMA sequence: bit-identical output.  Measured at one frame: 27 -> 19 us (profiles/r05_decode_zsplit.txt)
    int znb = 0;
    if (!bits_out && NPT <= 128 && (long long)B * G2 <= 128) {
        znb = ((long long)B * G2 <= 64) ? 1 : 2;
        if (NPT % (znb * 32) != 0 || NPT / (znb * 32) < 2) znb = 0;
    }
    for (int n0 = 0; n0 < NPT; n0 += 128) {
        const int nb = znb ? znb : ((NPT - n0 >= 128) ? 4 : (NPT - n0) / 32);
        const size_t lds = (size_t)2 * nb * 32 * (C + 8) * sizeof(_Float16) + (size_t)nb * 32 * sizeof(float);
        dim3 grid(G2, B, znb ? NPT / (znb * 32) : 1);
#ifdef VKN_DEBUG
        if (wide) {
#define D4_CASE(NBV)                                                                                                       \
    case NBV: {                                                                                                            \
        VKN_ALLOW_FULL_LDS(k_decode4<NBV>);                                                                                \
        hipLaunchKernelGGL((k_decode4<NBV>), grid, dim3(D4_THREADS), lds, stream, x, kfh, kfl, kb, out, N, NPT, n0, C, P, \
                           px_per_wg, xcd, fs);                                                                            \
    } break;
            switch (nb) {
                D4_CASE(1)
                D4_CASE(2)
                D4_CASE(3)
                D4_CASE(4)
                default:
                    return VKN_E_SHAPE;
            }
#undef D4_CASE
            VKN_CHECK_LAUNCH();
            continue;
        }
#endif
        dim3 block(DEC_THREADS);
#define DEC_LAUNCH_X(NBV, ABLV, RINGV, BITSV, OPTV, XHV)                                                       \
    do {                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_decode_mfma<NBV, ABLV, RINGV, BITSV, OPTV, XHV>));                               \
        hipLaunchKernelGGL((k_decode_mfma<NBV, ABLV, RINGV, BITSV, OPTV, XHV>), grid, block, lds, stream, x, kfh, kfl, kb, out, \
                           N, NPT, n0, C, P, px_per_wg, xcd, fs, bits_out, thr, (const float*)nullptr);        \
    } while (0)
#ifdef VKN_DEBUG
#define DEC_LAUNCH_XP() DEC_LAUNCH_X(4, 0, 3, 0, (DEC_OPT_DEFAULT | 8), 2)
#else
#define DEC_LAUNCH_XP() DEC_LAUNCH_X(4, 0, 3, 0, DEC_OPT_DEFAULT, 2)
#endif
    // half-storage x: the shipped variant only (no ablations, ring 3, default OPT)
#define DEC_LAUNCH_O(NBV, ABLV, RINGV, BITSV, OPTV)                                                            \
    do {                                                                                                       \
        if (xdt == 0) DEC_LAUNCH_X(NBV, ABLV, RINGV, BITSV, OPTV, 0);                                          \
        else if (ABLV != 0 || RINGV != 3 || OPTV != DEC_OPT_DEFAULT) return VKN_E_ARG;                         \
        else if (NBV == 4 && BITSV == 0 && xdt == 2 && vkn_dbg_env("VKN_DECODE_XPAIR", 0) != 0) DEC_LAUNCH_XP();  \
        else if (xdt == 1) DEC_LAUNCH_X(NBV, 0, 3, BITSV, DEC_OPT_DEFAULT, 1);                                 \
        else DEC_LAUNCH_X(NBV, 0, 3, BITSV, DEC_OPT_DEFAULT, 2);                                               \
    } while (0)
#ifdef VKN_DEBUG
        const int opt = vkn_dbg_env("VKN_DECODE_OPT", DEC_OPT_DEFAULT);
#define DEC_LAUNCH(NBV, ABLV, RINGV, BITSV)                                              \
    do {                                                                                 \
        if (ABLV != 0 || RINGV != 3 || BITSV != 0 || NBV != 4) DEC_LAUNCH_O(NBV, ABLV, RINGV, BITSV, DEC_OPT_DEFAULT); \
        else if (opt == 0) DEC_LAUNCH_O(4, 0, 3, 0, 0);                                  \
        else if (opt == 1) DEC_LAUNCH_O(4, 0, 3, 0, 1);                                  \
        else if (opt == 2) DEC_LAUNCH_O(4, 0, 3, 0, 2);                                  \
        else if (opt == 3) DEC_LAUNCH_O(4, 0, 3, 0, 3);                                  \
        else if (opt == 4) DEC_LAUNCH_O(4, 0, 3, 0, 4);                                  \
        else if (opt == 5) DEC_LAUNCH_O(4, 0, 3, 0, 5);                                  \
        else if (opt == 6) DEC_LAUNCH_O(4, 0, 3, 0, 6);                                  \
        else DEC_LAUNCH_O(4, 0, 3, 0, 7);                                                \
    } while (0)
#else
#define DEC_LAUNCH(NBV, ABLV, RINGV, BITSV) DEC_LAUNCH_O(NBV, ABLV, RINGV, BITSV, DEC_OPT_DEFAULT)
#endif
#ifdef VKN_DEBUG
        // time-attribution variants (WRONG results by construction) and the ring-depth A/B exist in the debug build only
        const int abl = vkn_dbg_env("VKN_DECODE_ABL", 0), ring = vkn_dbg_env("VKN_DECODE_RING", 3);
#define DEC_CASE(NBV)                                           \
    case NBV:                                                   \
        if (bits_out) DEC_LAUNCH(NBV, 0, 3, 1);                 \
        else if (abl == 0) {                                    \
            if (ring == 3) DEC_LAUNCH(NBV, 0, 3, 0);            \
            else DEC_LAUNCH(NBV, 0, 4, 0);                      \
        }                                                       \
        else if (NBV == 4 && abl == 1) DEC_LAUNCH(4, 1, 3, 0);  \
        else if (NBV == 4 && abl == 2) DEC_LAUNCH(4, 2, 3, 0);  \
        else if (NBV == 4 && abl == 3) DEC_LAUNCH(4, 3, 3, 0);  \
        else if (NBV == 4 && abl == 4) DEC_LAUNCH(4, 4, 3, 0);  \
        else if (NBV == 4 && abl == 5) DEC_LAUNCH(4, 5, 3, 0);  \
        else if (NBV == 4 && abl == 6) DEC_LAUNCH(4, 6, 3, 0);  \
        else if (NBV == 4 && abl == 7) DEC_LAUNCH(4, 7, 3, 0);  \
        else if (NBV == 4 && abl == 8) DEC_LAUNCH(4, 8, 3, 0);  \
        else if (NBV == 4 && abl == 9) DEC_LAUNCH(4, 9, 3, 0);  \
        else DEC_LAUNCH(NBV, 0, 3, 0);                          \
        break;
#else
#define DEC_CASE(NBV)                           \
    case NBV:                                   \
        if (bits_out) DEC_LAUNCH(NBV, 0, 3, 1); \
        else DEC_LAUNCH(NBV, 0, 3, 0);          \
        break;
#endif
        if (oscale) {   // outputs times a device scalar (training backward passes): the shipped variant of each storage type
#define DEC_LAUNCH_SC(NBV, XHV)                                                                                        \
    do {                                                                                                               \
        VKN_ALLOW_FULL_LDS((k_decode_mfma<NBV, 0, 3, 0, DEC_OPT_DEFAULT, XHV, 1>));                                    \
        hipLaunchKernelGGL((k_decode_mfma<NBV, 0, 3, 0, DEC_OPT_DEFAULT, XHV, 1>), grid, block, lds, stream, x, kfh, kfl, kb, out, \
                           N, NPT, n0, C, P, px_per_wg, xcd, fs, bits_out, thr, oscale);                               \
    } while (0)
#define DEC_CASE_SC(NBV)                          \
    case NBV:                                     \
        if (xdt == 0) DEC_LAUNCH_SC(NBV, 0);      \
        else if (xdt == 1) DEC_LAUNCH_SC(NBV, 1); \
        else DEC_LAUNCH_SC(NBV, 2);               \
        break;
            switch (nb) {
                DEC_CASE_SC(1)
                DEC_CASE_SC(2)
                DEC_CASE_SC(3)
                DEC_CASE_SC(4)
                default:
                    return VKN_E_SHAPE;
            }
#undef DEC_CASE_SC
#undef DEC_LAUNCH_SC
        } else
        switch (nb) {
            DEC_CASE(1)
            DEC_CASE(2)
            DEC_CASE(3)
            DEC_CASE(4)
            default:
                return VKN_E_SHAPE;
        }
#undef DEC_CASE
#undef DEC_LAUNCH
#undef DEC_LAUNCH_O
#undef DEC_LAUNCH_X
#undef DEC_LAUNCH_XP
        VKN_CHECK_LAUNCH();
    }
    return VKN_OK;
}

int vkn_launch_decode_ref(const float* x, const float* kern, const float* kb, float* out, int B, int N, int C, int P,
                          hipStream_t stream) {
    return vkn_launch_decode_ref_ex(x, kern, kb, out, B, N, C, P, 0, N, stream);
}

int vkn_launch_decode_ref_ex(const float* x, const float* kern, const float* kb, float* out, int B, int N, int C, int P,
                             int shared, int out_rows, hipStream_t stream) {
    if (out_rows < N) return VKN_E_ARG;
    const VknDecodeStrides fs{shared ? 0 : (long long)N * C, shared ? 0 : (long long)N, (long long)out_rows * P};
    dim3 grid((P + 255) / 256, N, B);
    hipLaunchKernelGGL(k_decode_ref, grid, dim3(256), 0, stream, x, kern, kb, out, N, C, P, fs);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_split_planes(const float* kern, _Float16* kfh, _Float16* kfl, int B, int N, int C, hipStream_t stream) {
    const int NPT = (N + 31) / 32 * 32;
    hipLaunchKernelGGL(k_split_planes, dim3(B * NPT), dim3(256), 0, stream, kern, kfh, kfl, N, NPT, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
