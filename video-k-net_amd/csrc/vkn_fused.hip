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

#ifdef VKN_DEBUG  // rejected / time-attribution variants live outside the product sources
#include "../../tools/experiments/fused_variants.inc"
#endif

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
