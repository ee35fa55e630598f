// vkn_fused.hip — mask decode of stage s FUSED with the mask gather of stage s + 1: ONE pass over x per stage boundary.
//
//   z[n][p]      = kb[n] + sum_c K[n][c] x[c][p]                    (stage s decode,  knet/det/kernel_update_head.py:247-260)
//   bit[n][p]    = z[n][p] >= thr_logit                             (stage s+1 binarise,                        :190-192)
//   xraw[n][c]   = sum_p bit[n][p] x[c][p],  cnt[n] = sum_p bit     (stage s+1 gather `einsum('bnhw,bchw->bnc')`, :195)
//
// The logits of an intermediate stage are consumed by nothing but this threshold, so neither they nor
// their bit words ever reach HBM, and x is streamed once where k_decode_mfma<bits> + k_gather_bits_w streamed it twice.
//
// Three variants were built and measured this round (cfg2, 32 frames per launch, 1.07 GB of x; unfused = k_decode_mfma<bits>
// 269 us + k_gather_bits_w 198 us + reduce 9 us = 476 us):
//   k_fused_dg   (debug lib)  4 waves, one per SIMD, kernels stationary in registers, 64-px tiles double-buffered      412-420 us
//   k_fused_dg8  (debug lib)  8 waves, all in the same phase, LO plane of the kernels in LDS                           416-424 us
//   k_fused_dgs  (RELEASE)    8 waves, role-specialised (4 decode + 4 gather, one of each per SIMD), 32-px tiles       402-407 us
// All three are FlashAttention-shaped (S = K x -> P = bit(S) -> O += P x^T per tile), produce bit-identical results, and are
// co-bound by dependent-MFMA issue and LDS / VALU latency, not by HBM (PMC, profiles/r02_pmc_fused.txt: matrix pipe 35 % busy,
// waves 32 % issue-stalled, 29 % waiting, VALU 25 %, LDS 28 %): 640 MFMA per 64 KB of x is ~140 us of matrix pipe per GB — the
// same order as the HBM time — and the f32 -> f16 hi/lo split costs 1.6 k VALU instructions per 64-px tile.  What the fusion buys
// today is one pass over x per stage boundary (6 -> 4 x-streaming kernels per frame) at 0.85x the time of the two kernels.
#include "vkn_common.h"
#include "vkn_launch.h"

#define FU_THREADS 256
#define FU_WAVES 4
#define FU_TILE 64

typedef unsigned int fu_u32x2 __attribute__((ext_vector_type(2)));

#ifdef VKN_DEBUG  // the two earlier variants: measured slower than k_fused_dgs, kept in the debug library for A/B (VKN_FUSED=0|1)
template <int NB, int C>
__global__ __launch_bounds__(FU_THREADS, 1) void k_fused_dg(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                            const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                            float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                            int N, int NPT, int n0, int P) {
    constexpr int KS = C / 16;                          // 16-channel k-steps of the decode contraction
    constexpr int NF = (KS + FU_WAVES - 1) / FU_WAVES;  // x fragments a wave loads per tile (k-steps w, w + 4, ...)
    constexpr int NCB = C / 32;                         // 32-channel blocks of the gather output
    constexpr int CBW = (NCB + FU_WAVES - 1) / FU_WAVES;
    constexpr int LDK = C + 8;                          // halfs per LDS row: (C + 8) * 2 B = odd multiple of 16 B
    constexpr int PLANE = FU_TILE * LDK;                // halfs of one plane of one tile image

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);  // [2 buffers][hi | lo][64 rows][LDK]
    half8* lut = reinterpret_cast<half8*>(dimg + 4 * PLANE);
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);  // [2 (even | odd pixels)][128 rows]
    float* kbs = reinterpret_cast<float*>(wbits + 256);        // [128]

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;

    // 64-px tiles s * G + gidx, s = 0, 1, ... (launcher: P % 64 == 0)
    const int nsup = ((P >> 6) - gidx + G - 1) / G;

    // (even nibble | odd nibble << 4) -> 8 halfs {0,1}: pixel e of an 8-px group = bit e/2 of the even (e even) / odd (e odd) nibble
    for (int v = tid; v < 256; v += FU_THREADS) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
        lut[v] = h;
    }
    if (tid < 128) {
        const int n = n0 + tid;
        kbs[tid] = (kb && tid < NB * 32 && n < N) ? kb[(size_t)b * N + n] : 0.f;
    }

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * C * P), 0, C * P * 4, 0x00020000);
    const int voff = (((g << 3) * P + 2 * li) << 2);  // lane (g, li): channels 8 g + e, pixels 2 li, 2 li + 1 of the tile

    // ---- x fragments of the next tile(s), raw fp32 in registers
    // NOTE on waits: vector loads return in order and the compiler's waitcnt insertion is exact only on branch-free code —
    // every load below is unconditional when KS divides evenly (ALLF), and `sched_barrier`s pin commit -> issue order, else
    // hipcc drains the prefetch (`s_waitcnt vmcnt(0)` right after the next tile's loads were issued: measured 2x slower).
    constexpr bool ALLF = (KS % FU_WAVES) == 0;
    fu_u32x2 raw[NF][8];
    auto issue = [&](int s, int f) {  // fragment f of tile s: k-step wave + 4 f
        const int ks = wave + FU_WAVES * f;
        if (ALLF || ks < KS) {  // uniform
            const int p0 = (s * G + gidx) << 6;
            const int soff = ((ks << 4) * P + p0) << 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
        }
    };
    // split fragment f and write it to tile image `buf`: rows li (even pixel) and 32 + li (odd pixel), columns 16 ks + 8 g ..
    auto commit = [&](int buf, int f) {
        const int ks = wave + FU_WAVES * f;
        if (ALLF || ks < KS) {
            half8 h0, l0, h1, l1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned u0 = raw[f][e][0], u1 = raw[f][e][1];
                _Float16 h, l;
                vkn_split_f16(__uint_as_float(u0), h, l);
                h0[e] = h;
                l0[e] = l;
                vkn_split_f16(__uint_as_float(u1), h, l);
                h1[e] = h;
                l1[e] = l;
            }
            _Float16* dh = dimg + (size_t)buf * 2 * PLANE + li * LDK + (ks << 4) + (g << 3);
            *reinterpret_cast<half8*>(dh) = h0;
            *reinterpret_cast<half8*>(dh + PLANE) = l0;
            *reinterpret_cast<half8*>(dh + 32 * LDK) = h1;
            *reinterpret_cast<half8*>(dh + 32 * LDK + PLANE) = l1;
        }
    };

#pragma unroll
    for (int f = 0; f < NF; ++f) issue(0, f);  // (tile gidx < P / 64 exists even when nsup == 0 only if G <= P / 64: launcher)

    // ---- decode operand A: this wave's 32 kernel rows, stationary in registers
    half8 Ah[KS], Al[KS];
    {
        const int n = n0 + wave * 32 + li;
        const bool ok = (wave < NB) && (n < N);  // rows >= N of the planes are never written by the producer: zero
        const size_t base = ((size_t)b * NPT + (ok ? n : 0)) * C + (g << 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) {
                vh = *reinterpret_cast<const half8*>(kfh + base + (ks << 4));
                vl = *reinterpret_cast<const half8*>(kfl + base + (ks << 4));
            }
            Ah[ks] = vh;
            Al[ks] = vl;
        }
    }

    f32x16 accg[CBW][NB];
#pragma unroll
    for (int j = 0; j < CBW; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accg[j][nb][r] = 0.f;
    unsigned cnt_i = 0;  // lanes 0..31 of wave nb: ON pixels of row 32 nb + lane

    {
        const int last = max(nsup - 1, 0);  // (a workgroup without tiles converts / re-reads tile 0 of its frame: harmless)
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(0, f);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), UNCONDITIONALLY on the way into the loop: the prologue's loads (kernel
        __builtin_amdgcn_sched_barrier(0);   // rows, first tile) are complete — inside the loop the only vector loads in flight
#pragma unroll                               // are the next tile's, which no phase touches
        for (int f = 0; f < NF; ++f) issue(min(1, last), f);
    }
    __syncthreads();

    for (int s = 0; s < nsup; ++s) {
        const int buf = s & 1;
        const _Float16* dh = dimg + (size_t)buf * 2 * PLANE;
        // ---------------- decode phase: wave = n-block
        if (wave < NB) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float k0 = kbs[wave * 32 + vkn_cd_row(r, lane)];
                acc0[r] = k0;
                acc1[r] = k0;
            }
            const _Float16* bp = dh + li * LDK + (g << 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 bh0 = *reinterpret_cast<const half8*>(bp + (ks << 4));
                const half8 bl0 = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                const half8 bh1 = *reinterpret_cast<const half8*>(bp + (ks << 4) + 32 * LDK);
                const half8 bl1 = *reinterpret_cast<const half8*>(bp + (ks << 4) + 32 * LDK + PLANE);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bh0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bh1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bl0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bl1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], bh0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], bh1, acc1, 0, 0, 0);
            }
            // bit words of this wave's 32 rows: lane = row
            int we = 0, wo = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long long m0 = __ballot(acc0[r] >= thr);  // even pixels: bit 32 g' + li'
                const unsigned long long m1 = __ballot(acc1[r] >= thr);  // odd pixels
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * g2;  // compile-time
                    const bool mine = lane == row;
                    we = mine ? (int)(unsigned)(m0 >> (32 * g2)) : we;
                    wo = mine ? (int)(unsigned)(m1 >> (32 * g2)) : wo;
                }
                __builtin_amdgcn_sched_barrier(0);  // keep each pair of ballots next to its selects (else the masks spill SGPRs)
            }
            if (lane < 32) {
                wbits[wave * 32 + lane] = (unsigned)we;
                wbits[128 + wave * 32 + lane] = (unsigned)wo;
                cnt_i += __popc((unsigned)we) + __popc((unsigned)wo);
            }
        }
        __syncthreads();  // A: the tile's bit words are visible; every wave has left the previous tile's gather phase

        // ---------------- gather phase: wave = channel blocks wave + 4 j; the next tile is converted between the MFMA groups
        unsigned wev[NB], wov[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            wev[nb] = wbits[nb * 32 + li];
            wov[nb] = wbits[128 + nb * 32 + li];
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int bsh = 8 * ps + 4 * g;  // pixels 16 ps + 8 g + e  <->  bit bsh + e / 2 of the even (e even) / odd word
            half8 a[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = lut[((wev[nb] >> bsh) & 0xFu) | (((wov[nb] >> bsh) & 0xFu) << 4)];
#pragma unroll
            for (int j = 0; j < CBW; ++j) {
                const int cb = wave + FU_WAVES * j;
                if (cb < NCB) {  // uniform
                    // pixel 16 ps + 8 g + e lives in row (8 ps + 4 g + e / 2) + 32 (e & 1)
                    const _Float16* cp = dh + (8 * ps + 4 * g) * LDK + cb * 32 + li;
                    half8 bh, bl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bh[e] = cp[((e >> 1) + 32 * (e & 1)) * LDK];
                        bl[e] = cp[((e >> 1) + 32 * (e & 1)) * LDK + PLANE];
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb], bh, accg[j][nb], 0, 0, 0);
                        accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb], bl, accg[j][nb], 0, 0, 0);
                    }
                }
            }
            if (ps < NF) {  // next tile: convert fragment ps (in flight since the previous tile), then request the tile after
                __builtin_amdgcn_sched_barrier(0);
                commit(buf ^ 1, ps);  // (past the last tile: a harmless re-conversion of the clamped re-read)
                __builtin_amdgcn_sched_barrier(0);
                issue(min(s + 2, nsup - 1), ps);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // B: the next tile's image is complete; this tile's image and bit words are free
    }

    // ---- this workgroup's partial (the layout of k_gather_mfma / k_gather_bits_w; rows of the n-chunk)
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;
#pragma unroll
    for (int j = 0; j < CBW; ++j) {
        const int cb = wave + FU_WAVES * j;
        if (cb < NCB) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                    pp[(size_t)n * C + cb * 32 + li] = accg[j][nb][r];
                }
        }
    }
    if (wave < NB && lane < 32) cntp[((size_t)b * G + gidx) * NPT + n0 + wave * 32 + lane] = (float)cnt_i;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_fused_dg8 — the same pass with TWO waves per SIMD (512 threads): measured on MI355X, the one-wave-per-SIMD kernel above
// leaves every LDS round trip of its single instruction stream exposed (per tile ~15 k cycles for 5.1 k cycles of MFMA), so it
// only ties with the two-kernel path.  With a partner wave on each SIMD the hardware overlaps one wave's LDS / VALU phases with
// the other's MFMAs.  256 registers per lane instead of 512 -> only the HI plane of the wave's kernel rows stays in registers
// (C / 4 registers); the LO plane lives in LDS (67.6 KB) and the x tile image is single-buffered (67.6 KB):
//   decode role of wave w: n-block w & 3, strip w >> 2 (even / odd pixels)      48 MFMA per tile
//   gather role of wave w: channel block w (all n-blocks)                       8 NB MFMA per tile
//   loads: k-steps w, w + 8 of the NEXT tile in flight in registers while this tile is multiplied
// Three workgroup barriers per tile (image complete / bit words complete / image free).  Results: bit-identical to k_fused_dg.
#define FU8_THREADS 512
#define FU8_WAVES 8

template <int NB, int C>
__global__ __launch_bounds__(FU8_THREADS, 2) void k_fused_dg8(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                              const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                              float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                              int N, int NPT, int n0, int P) {
    constexpr int KS = C / 16;
    constexpr int NF = (KS + FU8_WAVES - 1) / FU8_WAVES;
    constexpr int NCB = C / 32;
    constexpr int LDK = C + 8;
    constexpr int PLANE = FU_TILE * LDK;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);          // [hi | lo][64 rows][LDK]: the x tile
    _Float16* alo = dimg + 2 * PLANE;                            // [128 rows][LDK]: LO plane of the frame's kernels
    half8* lut = reinterpret_cast<half8*>(alo + 128 * LDK);
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);    // [2 (even | odd pixels)][128 rows]
    float* kbs = reinterpret_cast<float*>(wbits + 256);          // [128]
    unsigned* cnt2 = reinterpret_cast<unsigned*>(kbs + 128);     // [128]: odd-pixel counts, added by the even-pixel waves at the end

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int dnb = wave & 3, dt = wave >> 2;  // decode role
    const bool has_dec = dnb < NB;
    const bool has_cb = wave < NCB;

    const int nsup = ((P >> 6) - gidx + G - 1) / G;

    for (int v = tid; v < 256; v += FU8_THREADS) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
        lut[v] = h;
    }
    if (tid < 128) {
        const int n = n0 + tid;
        kbs[tid] = (kb && tid < NB * 32 && n < N) ? kb[(size_t)b * N + n] : 0.f;
        cnt2[tid] = 0u;
    }
    {  // LO plane of the chunk's kernel rows -> LDS (rows >= N: zero)
        const _Float16* gl = kfl + ((size_t)b * NPT + n0) * C;
        constexpr int cpr = C >> 3;
        for (int i = tid; i < NB * 32 * cpr; i += FU8_THREADS) {
            const int r = i / cpr, q = i - r * cpr;
            half8 vl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n0 + r < N) vl = *reinterpret_cast<const half8*>(gl + (size_t)r * C + q * 8);
            *reinterpret_cast<half8*>(alo + r * LDK + q * 8) = vl;
        }
    }

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * C * P), 0, C * P * 4, 0x00020000);
    const int voff = (((g << 3) * P + 2 * li) << 2);

    constexpr bool ALLF = (KS % FU8_WAVES) == 0;  // see the note on waits in k_fused_dg
    fu_u32x2 raw[NF][8];
    auto issue = [&](int s, int f) {
        const int ks = wave + FU8_WAVES * f;
        if (ALLF || ks < KS) {
            const int p0 = (s * G + gidx) << 6;
            const int soff = ((ks << 4) * P + p0) << 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
        }
    };
    auto commit = [&](int f) {
        const int ks = wave + FU8_WAVES * f;
        if (ALLF || ks < KS) {
            half8 h0, l0, h1, l1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned u0 = raw[f][e][0], u1 = raw[f][e][1];
                _Float16 h, l;
                vkn_split_f16(__uint_as_float(u0), h, l);
                h0[e] = h;
                l0[e] = l;
                vkn_split_f16(__uint_as_float(u1), h, l);
                h1[e] = h;
                l1[e] = l;
            }
            _Float16* dh = dimg + li * LDK + (ks << 4) + (g << 3);
            *reinterpret_cast<half8*>(dh) = h0;
            *reinterpret_cast<half8*>(dh + PLANE) = l0;
            *reinterpret_cast<half8*>(dh + 32 * LDK) = h1;
            *reinterpret_cast<half8*>(dh + 32 * LDK + PLANE) = l1;
        }
    };

#pragma unroll
    for (int f = 0; f < NF; ++f) issue(0, f);

    // HI plane of this wave's 32 kernel rows: stationary in registers
    half8 Ah[KS];
    {
        const int n = n0 + dnb * 32 + li;
        const bool ok = has_dec && (n < N);
        const size_t base = ((size_t)b * NPT + (ok ? n : 0)) * C + (g << 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 vh = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) vh = *reinterpret_cast<const half8*>(kfh + base + (ks << 4));
            Ah[ks] = vh;
        }
    }

    f32x16 accg[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[nb][r] = 0.f;
    unsigned cnt_i = 0;

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): prologue loads complete (see k_fused_dg) — also the first tile's fragments
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < nsup; ++s) {
        // ---------------- the tile's image: split this wave's fragments (requested one tile ago), THEN request the next tile's
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(min(s + 1, nsup - 1), f);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // X: image complete (and, first tile, LO plane / table / bias staged)

        // ---------------- decode phase: wave = (n-block, strip)
        if (has_dec) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = kbs[dnb * 32 + vkn_cd_row(r, lane)];
            const _Float16* bp = dimg + (32 * dt + li) * LDK + (g << 3);
            const _Float16* ap = alo + (dnb * 32 + li) * LDK + (g << 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 bh = *reinterpret_cast<const half8*>(bp + (ks << 4));
                const half8 bl = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                const half8 al = *reinterpret_cast<const half8*>(ap + (ks << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            }
            int wd = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long long m = __ballot(acc[r] >= thr);
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * g2;
                    wd = (lane == row) ? (int)(unsigned)(m >> (32 * g2)) : wd;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (lane < 32) {
                wbits[dt * 128 + dnb * 32 + lane] = (unsigned)wd;
                cnt_i += __popc((unsigned)wd);
            }
        }
        __syncthreads();  // A: bit words complete

        // ---------------- gather phase: wave = channel block
        if (has_cb) {
            unsigned wev[NB], wov[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                wev[nb] = wbits[nb * 32 + li];
                wov[nb] = wbits[128 + nb * 32 + li];
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int bsh = 8 * ps + 4 * g;
                const _Float16* cp = dimg + (8 * ps + 4 * g) * LDK + wave * 32 + li;
                half8 bh, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bh[e] = cp[((e >> 1) + 32 * (e & 1)) * LDK];
                    bl[e] = cp[((e >> 1) + 32 * (e & 1)) * LDK + PLANE];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const half8 a = lut[((wev[nb] >> bsh) & 0xFu) | (((wov[nb] >> bsh) & 0xFu) << 4)];
                    accg[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, accg[nb], 0, 0, 0);
                    accg[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, accg[nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // B: image and bit words are free
    }

    float* pp = part + ((size_t)b * G + gidx) * NPT * C;
    if (has_cb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                pp[(size_t)n * C + wave * 32 + li] = accg[nb][r];
            }
    }
    // ON-pixel counts: the odd-pixel waves hand theirs over through LDS (integers: exact in any order)
    if (has_dec && dt == 1 && lane < 32) cnt2[dnb * 32 + lane] = cnt_i;
    __syncthreads();
    if (has_dec && dt == 0 && lane < 32)
        cntp[((size_t)b * G + gidx) * NPT + n0 + dnb * 32 + lane] = (float)(cnt_i + cnt2[dnb * 32 + lane]);
}


#endif  // VKN_DEBUG (k_fused_dg, k_fused_dg8)

// ---------------------------------------------------------------------------------------------------------------------------
// k_fused_dgs — the same pass with ROLE-SPECIALISED waves.  Measured (tools/perf_r02.py, cfg2, 32 frames): k_fused_dg 420 us,
// k_fused_dg8 416 us for 137 us of matrix pipe: both walk every tile through convert -> decode -> gather phases separated by
// workgroup barriers, so at any moment all waves of a CU are in the SAME phase and the LDS / VALU / matrix pipes take turns
// (per tile ~4.4 k LDS + ~4 k VALU + 5.1 k MFMA cycles, measured 13.6 k: no overlap at all).  Here the phases run CONCURRENTLY
// on different waves of each SIMD:
//   waves 0-3 (one per SIMD): decode role, n-block = wave; kernel rows (hi + lo) stationary in registers; tile i
//   waves 4-7 (one per SIMD): gather role, channel blocks wave - 4 and wave; tile i - 1 (its bit words are ready)
//   all waves: load / split / write 1/8 of tile i + 1 into the third image buffer, request tile i + 2
// so each SIMD always has one wave feeding MFMAs from LDS reads while its partner does ballots / conversions / u16 reads.
// Tiles are 32 px (one MFMA strip): three image buffers fit (3 x 33.8 KB), ONE workgroup barrier per tile.  The 32-px tiles are
// walked in the order of the 64-px super-tiles' halves, the MFMA k index maps to the same pixels, so partials and results stay
// bit-identical to the other variants and to the unfused path.
#define FS_THREADS 512
#define FS_TILE 32

// XH (x storage): 0 = fp32; 1 = fp16, 2 = bf16 (converted to f16): the loader's pixel pair is one dword, only the hi plane of the
// tile image is written / read and every MFMA against x_lo disappears (decode 3 -> 2, gather 2 -> 1 per operand pair).  On
// x' = float(half(x)) the fp32 kernel returns the same bits.
template <int NB, int C, int XH = 0>
__global__ __launch_bounds__(FS_THREADS, 2) void k_fused_dgs(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                              const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                              float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                              int N, int NPT, int n0, int P) {
    constexpr int KS = C / 16;
    constexpr int NF = (KS + 7) / 8;             // 16-channel fragments a wave loads per tile (k-steps wave, wave + 8)
    constexpr bool ALLF = (KS % 8) == 0;
    constexpr int NCB = C / 32;
    constexpr int CBW = (NCB + 3) / 4;           // channel blocks per gather wave
    constexpr int LDK = C + 8;
    constexpr int PLANE = FS_TILE * LDK;         // halfs of one plane of one tile image
    constexpr int IMG = 2 * PLANE;               // hi | lo

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);         // [3 buffers][hi | lo][32 px][LDK]
    half8* lut = reinterpret_cast<half8*>(dimg + 3 * IMG);      // (even nibble | odd nibble << 4) -> 8 halfs {0,1}
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);   // [2 buffers][128 rows]: bit i = image row i of the tile
    float* kbs = reinterpret_cast<float*>(wbits + 256);         // [128]

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;

    const int nsup = ((P >> 6) - gidx + G - 1) / G;
    const int T = 2 * nsup;  // 32-px tiles: halves of the 64-px super-tiles s * G + gidx
    auto tile_p0 = [&](int t) { return (((t >> 1) * G + gidx) << 6) + ((t & 1) << 5); };

    // image row of pixel p: (p >> 1) + 16 (p & 1) — even pixels in rows 0..15, odd pixels in rows 16..31 (the loader's lanes own
    // pixel PAIRS; this keeps its 8-byte LDS writes at a 528-byte lane stride).  MFMA column / ballot bit i of the decode is
    // therefore pixel 2 i (i < 16) or 2 (i - 16) + 1, and the gather's table is the (even nibble | odd nibble << 4) one.
    for (int v = tid; v < 256; v += FS_THREADS) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
        lut[v] = h;
    }
    if (tid < 128) {
        const int n = n0 + tid;
        kbs[tid] = (kb && tid < NB * 32 && n < N) ? kb[(size_t)b * N + n] : 0.f;
    }

    const __amdgpu_buffer_rsrc_t xrs =
        XH ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P), 0,
                                               C * P * 2, 0x00020000)
           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * C * P), 0, C * P * 4, 0x00020000);
    // loader: lane (q, lp) = (lane >> 4, lane & 15): channels 4 q + e (e = 0..3) of the 16-channel fragment, pixels 2 lp, 2 lp + 1
    const int lq = lane >> 4, lp = lane & 15;
    constexpr int XSH = XH ? 1 : 2;  // log2(bytes per stored element)
    const int voff = (((lq << 2) * P + 2 * lp) << XSH);
    fu_u32x2 raw[NF][4];
    auto issue = [&](int t, int f) {
        const int ks = wave + 8 * f;
        if (ALLF || ks < KS) {
            const int soff = ((ks << 4) * P + tile_p0(t)) << XSH;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // half storage: a 32-px tile is HALF a 128-byte line of its channel row; the other half is the next tile of this same
                // workgroup.  With the streaming hint (sc0 | nt) the line was fetched from HBM twice (PMC FETCH_SIZE = 2x the x bytes,
                // profiles/r02n); default caching keeps it for the second half.
                if (XH) raw[f][e] = fu_u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, soff + ((e * P) << 1), 0), 0u};
                else raw[f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
            }
        }
    };
    auto commit = [&](int buf, int f) {  // rows lp (pixel 2 lp) and 16 + lp (pixel 2 lp + 1), columns 16 ks + 4 q .. + 4
        const int ks = wave + 8 * f;
        if (ALLF || ks < KS) {
            half4 h0, l0, h1, l1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u0 = raw[f][e][0], u1 = raw[f][e][1];
                _Float16 h, l;
                if (XH == 1) {  // fp16 pair: low half = even pixel
                    h0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 & 0xFFFFu));
                    h1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 >> 16));
                } else if (XH == 2) {  // bf16 pair -> fp32 (exact) -> f16
                    h0[e] = (_Float16)__uint_as_float(u0 << 16);
                    h1[e] = (_Float16)__uint_as_float(u0 & 0xFFFF0000u);
                } else {
                    vkn_split_f16(__uint_as_float(u0), h, l);
                    h0[e] = h;
                    l0[e] = l;
                    vkn_split_f16(__uint_as_float(u1), h, l);
                    h1[e] = h;
                    l1[e] = l;
                }
            }
            _Float16* dh = dimg + (size_t)buf * IMG + lp * LDK + (ks << 4) + (lq << 2);
            *reinterpret_cast<half4*>(dh) = h0;
            *reinterpret_cast<half4*>(dh + 16 * LDK) = h1;
            if (!XH) {
                *reinterpret_cast<half4*>(dh + PLANE) = l0;
                *reinterpret_cast<half4*>(dh + 16 * LDK + PLANE) = l1;
            }
        }
    };

#pragma unroll
    for (int f = 0; f < NF; ++f) issue(0, f);

    float* pp = part + ((size_t)b * G + gidx) * NPT * C;
    const int tlast = max(T - 1, 0);

    if (wave < 4) {
        // =============================================================== decode role: n-block `wave`
        const bool has_dec = wave < NB;
        half8 Ah[KS], Al[KS];
        {
            const int n = n0 + wave * 32 + li;
            const bool ok = has_dec && (n < N);
            const size_t base = ((size_t)b * NPT + (ok ? n : 0)) * C + (g << 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
                    vh = *reinterpret_cast<const half8*>(kfh + base + (ks << 4));
                    vl = *reinterpret_cast<const half8*>(kfl + base + (ks << 4));
                }
                Ah[ks] = vh;
                Al[ks] = vl;
            }
        }
        unsigned cnt_i = 0;
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(0, f);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): prologue loads complete (see the note on waits in k_fused_dg)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(min(1, tlast), f);
        __syncthreads();
        for (int i = 0; i <= T; ++i) {
            if (has_dec && i < T) {
                const _Float16* bp = dimg + (size_t)(i % 3) * IMG + li * LDK + (g << 3);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = kbs[wave * 32 + vkn_cd_row(r, lane)];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 bh = *reinterpret_cast<const half8*>(bp + (ks << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bh, acc, 0, 0, 0);
                    if (!XH) {
                        const half8 bl = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], bl, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], bh, acc, 0, 0, 0);
                }
                int wd = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long m = __ballot(acc[r] >= thr);  // bit 32 g' + li' : row (r) + 4 g', image row li'
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * g2;
                        wd = (lane == row) ? (int)(unsigned)(m >> (32 * g2)) : wd;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (lane < 32) {
                    wbits[(i & 1) * 128 + wave * 32 + lane] = (unsigned)wd;
                    cnt_i += __popc((unsigned)wd);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) commit((i + 1) % 3, f);  // tile i + 1 (clamped re-reads past the end: harmless)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) issue(min(i + 2, tlast), f);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
        if (has_dec && lane < 32) cntp[((size_t)b * G + gidx) * NPT + n0 + wave * 32 + lane] = (float)cnt_i;
    } else {
        // =============================================================== gather role: channel blocks wave - 4 (+ 4)
        const int gw = wave - 4;
        f32x16 accg[CBW][NB];
#pragma unroll
        for (int j = 0; j < CBW; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[j][nb][r] = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(0, f);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(min(1, tlast), f);
        __syncthreads();
        for (int i = 0; i <= T; ++i) {
            if (i >= 1) {
                const int t = i - 1;
                const _Float16* dh = dimg + (size_t)(t % 3) * IMG;
                unsigned wv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wv[nb] = wbits[(t & 1) * 128 + nb * 32 + li];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    half8 a[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)  // pixels 16 ps + 8 g + e: even e -> bit 8 ps + 4 g + e / 2, odd e -> 16 + the same
                        a[nb] = lut[((wv[nb] >> (8 * ps + 4 * g)) & 0xFu) | (((wv[nb] >> (16 + 8 * ps + 4 * g)) & 0xFu) << 4)];
#pragma unroll
                    for (int j = 0; j < CBW; ++j) {
                        const int cb = gw + 4 * j;
                        if (cb < NCB) {
                            const _Float16* cp = dh + (8 * ps + 4 * g) * LDK + cb * 32 + li;
                            half8 bh, bl;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                bh[e] = cp[((e >> 1) + 16 * (e & 1)) * LDK];
                                if (!XH) bl[e] = cp[((e >> 1) + 16 * (e & 1)) * LDK + PLANE];
                            }
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb], bh, accg[j][nb], 0, 0, 0);
                                if (!XH) accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nb], bl, accg[j][nb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) commit((i + 1) % 3, f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) issue(min(i + 2, tlast), f);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < CBW; ++j) {
            const int cb = gw + 4 * j;
            if (cb < NCB) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                        pp[(size_t)n * C + cb * 32 + li] = accg[j][nb][r];
                    }
            }
        }
    }
}

static size_t fuseds_lds_bytes(int C) { return (size_t)3 * 2 * FS_TILE * (C + 8) * sizeof(_Float16) + 256 * 16 + 256 * 4 + 128 * 4; }

#ifdef VKN_DEBUG
static size_t fused_lds_bytes(int C) { return (size_t)4 * FU_TILE * (C + 8) * sizeof(_Float16) + 256 * 16 + 256 * 4 + 128 * 4; }
static size_t fused8_lds_bytes(int C) {
    return (size_t)(2 * FU_TILE + 128) * (C + 8) * sizeof(_Float16) + 256 * 16 + 256 * 4 + 128 * 4 + 128 * 4;
}
#endif

int vkn_fused_supported(int C, int P) {
    return (C == 64 || C == 128 || C == 256) && (P % 64) == 0 && (size_t)C * P * 4 < ((size_t)1 << 31);
}

// Stage s decode (kernels kfh / kfl [B][NPT][C] f16 planes, bias kb [B][N]) fused with the stage s + 1 gather: xraw [B][N][C],
// cnt [B][N]; part / cntp: the gather's workspace ([B][G][NPT][C], [B][G][NPT], G = vkn_gather_groups(B, P)).
int vkn_launch_fused_decode_gather(const float* x, const _Float16* kfh, const _Float16* kfl, const float* kb, float thr,
                                   float* xraw, float* cnt, float* part, float* cntp, int B, int N, int C, int P,
                                   hipStream_t stream, int xdt) {
    if (B <= 0 || N <= 0 || P <= 0 || xdt < 0 || xdt > 2) return VKN_E_ARG;
    if (!vkn_fused_supported(C, P)) return VKN_E_SHAPE;
    const int NPT = (N + 31) / 32 * 32;
    const int G = vkn_gather_groups(B, P);
#ifdef VKN_DEBUG
    const int variant = vkn_dbg_env("VKN_FUSED", 2);  // debug build A/B: 0 = k_fused_dg, 1 = k_fused_dg8, 2 = k_fused_dgs
    const bool eight = variant == 1;
    const size_t lds = variant == 2 ? fuseds_lds_bytes(C) : (eight ? fused8_lds_bytes(C) : fused_lds_bytes(C));
#else
    const size_t lds = fuseds_lds_bytes(C);
#endif
    for (int n0 = 0; n0 < NPT; n0 += 128) {
        const int nb = (NPT - n0 >= 128) ? 4 : (NPT - n0) / 32;
        dim3 grid(G, B, 1);
#define FU_LAUNCH_X(NBV, CV, XHV)                                                                                              \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_dgs<NBV, CV, XHV>));                                                                       \
        hipLaunchKernelGGL((k_fused_dgs<NBV, CV, XHV>), grid, dim3(FS_THREADS), lds, stream, x, kfh, kfl, kb, thr, part, cntp, \
                           N, NPT, n0, P);                                                                                     \
    } while (0)
#define FU_LAUNCH_S(NBV, CV)                        \
    do {                                            \
        if (xdt == 1) FU_LAUNCH_X(NBV, CV, 1);      \
        else if (xdt == 2) FU_LAUNCH_X(NBV, CV, 2); \
        else FU_LAUNCH_X(NBV, CV, 0);               \
    } while (0)
#ifdef VKN_DEBUG
#define FU_LAUNCH(NBV, CV)                                                                                                     \
    do {                                                                                                                       \
        if (variant == 2 || xdt != 0) FU_LAUNCH_S(NBV, CV);                                                                    \
        else if (eight) {                                                                                                      \
            VKN_ALLOW_FULL_LDS((k_fused_dg8<NBV, CV>));                                                                        \
            hipLaunchKernelGGL((k_fused_dg8<NBV, CV>), grid, dim3(FU8_THREADS), lds, stream, x, kfh, kfl, kb, thr, part, cntp, \
                               N, NPT, n0, P);                                                                                 \
        } else {                                                                                                               \
            VKN_ALLOW_FULL_LDS((k_fused_dg<NBV, CV>));                                                                         \
            hipLaunchKernelGGL((k_fused_dg<NBV, CV>), grid, dim3(FU_THREADS), lds, stream, x, kfh, kfl, kb, thr, part, cntp,   \
                               N, NPT, n0, P);                                                                                 \
        }                                                                                                                      \
    } while (0)
#else
#define FU_LAUNCH(NBV, CV) FU_LAUNCH_S(NBV, CV)
#endif
#define FU_CASE(NBV)                               \
    case NBV:                                      \
        if (C == 256) FU_LAUNCH(NBV, 256);         \
        else if (C == 128) FU_LAUNCH(NBV, 128);    \
        else FU_LAUNCH(NBV, 64);                   \
        break;
        switch (nb) {
            FU_CASE(1)
            FU_CASE(2)
            FU_CASE(3)
            FU_CASE(4)
            default:
                return VKN_E_SHAPE;
        }
#undef FU_CASE
#undef FU_LAUNCH
#undef FU_LAUNCH_S
#undef FU_LAUNCH_X
        VKN_CHECK_LAUNCH();
    }
    return vkn_launch_gather_reduce(part, cntp, xraw, cnt, B, N, C, G, stream);  // the unfused path's fixed-order second pass
}
