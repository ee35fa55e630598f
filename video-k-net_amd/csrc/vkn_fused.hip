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
// V (variant bits, A/B in the debug library; the release library instantiates FS_V_DEFAULT only — all bit-identical):
//   1  decode: x fragments of k-step ks + 2 are requested BEFORE the MFMAs of k-step ks (3-deep register ring; r02: the four
//      reads of a k-step pair were issued right in front of their MFMAs and waited for — an LDS round trip exposed 8x per tile)
//   2  decode: ballot words go to "lane = row" through v_writelane (16 compares + 32 writelanes) instead of 32 per-lane compare
//      masks held in 64 SGPRs (28 of them spilled) and a chain of 64 dependent v_cndmask
//   4  gather: the B fragment (8 pixels of one channel = a COLUMN of the [pixel][channel] image) comes from two
//      ds_read_b64_tr_b16 transpose reads instead of eight ds_read_u16 + four v_perm, and every LDS operand of the tile (bit words ->
//      table rows, 16 transposed fragments) is requested up front instead of stage by stage
#define FS_V_DEFAULT 7
//   8  (debug library only) s_memtime stamps around the phases of every wave of workgroup (0, 0): cycles per tile spent in
//      [role work | waiting for the next tile's loads | split + LDS writes | load issue | barrier], read back by vkn_dbg_fused_prof
#ifdef VKN_DEBUG
__device__ unsigned long long g_fs_prof[8][8];
#define FS_STAMP(k)                                                     \
    do {                                                                \
        if constexpr ((V & 8) != 0) {                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            prof[k] += now_ - tprev;                                    \
            tprev = now_;                                               \
        }                                                               \
    } while (0)
#else
#define FS_STAMP(k) do { } while (0)
#endif
typedef short fs_short4 __attribute__((ext_vector_type(4)));
typedef short fs_short8 __attribute__((ext_vector_type(8)));
// lane `row` of `wd` <- the wave-uniform `val` (no builtin in this hipcc; the lane select is an immediate: no SGPR hazard)
#define FS_WRITELANE(wd, val, row) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(wd) : "s"(val), "n"(row))

template <int NB, int C, int XH = 0, int V = FS_V_DEFAULT>
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
    unsigned long long prof[5] = {0, 0, 0, 0, 0}, tprev = 0;
    (void)prof; (void)tprev;

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
#ifdef VKN_DEBUG
        if constexpr ((V & 8) != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i = 0; i <= T; ++i) {
            if (has_dec && i < T) {
                const _Float16* bp = dimg + (size_t)(i % 3) * IMG + li * LDK + (g << 3);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = kbs[wave * 32 + vkn_cd_row(r, lane)];
                if constexpr (V & 1) {
                    half8 rbh[3], rbl[3];
                    auto ldb = [&](int slot, int ks) {
                        rbh[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4));
                        if (!XH) rbl[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                    };
                    ldb(0, 0);
                    if (KS > 1) ldb(1, 1);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) ldb((ks + 2) % 3, ks + 2);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbh[ks % 3], acc, 0, 0, 0);
                        if (!XH) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbl[ks % 3], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], rbh[ks % 3], acc, 0, 0, 0);
                    }
                    // pin the pipeline (else the scheduler sinks every read back in front of its use and the ring collapses):
                    // [bias rows: 4 reads][k-steps 0, 1][k-step 2 | MFMAs 0][k-step 3 | MFMAs 1] ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (KS > 1 ? 2 : 1) * (XH ? 1 : 2), 0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, XH ? 1 : 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, XH ? 2 : 3, 0);
                    }
                } else {
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
                }
                int wd = 0;
                if constexpr (V & 2) {
                    // all 16 compares first, the writelanes behind a scheduling barrier: v_writelane right behind the v_cmp that wrote
                    // its SGPR operand reads a stale value (measured; hipcc pads nothing inside an asm statement)
                    unsigned long long m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] >= thr);  // bit 32 g' + li' : row (r) + 4 g', image row li'
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);                   // compile-time
                        const int mlo = (int)(unsigned)m[r], mhi = (int)(unsigned)(m[r] >> 32);
                        FS_WRITELANE(wd, mlo, row);
                        FS_WRITELANE(wd, mhi, row + 4);
                    }
                } else {
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
                }
                if (lane < 32) {
                    wbits[(i & 1) * 128 + wave * 32 + lane] = (unsigned)wd;
                    cnt_i += __popc((unsigned)wd);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(0);
            if constexpr ((V & 8) != 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_sched_barrier(0);
            }
            FS_STAMP(1);
#pragma unroll
            for (int f = 0; f < NF; ++f) commit((i + 1) % 3, f);  // tile i + 1 (clamped re-reads past the end: harmless)
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(2);
#pragma unroll
            for (int f = 0; f < NF; ++f) issue(min(i + 2, tlast), f);
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(3);
            __syncthreads();
            FS_STAMP(4);
        }
#ifdef VKN_DEBUG
        if constexpr ((V & 8) != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 5; ++k) g_fs_prof[wave][k] = prof[k];
#endif
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
#ifdef VKN_DEBUG
        if constexpr ((V & 8) != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i = 0; i <= T; ++i) {
            if (i >= 1) {
                const int t = i - 1;
                const _Float16* dh = dimg + (size_t)(t % 3) * IMG;
                unsigned wv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wv[nb] = wbits[(t & 1) * 128 + nb * 32 + li];
                if constexpr (V & 4) {
                    // B fragment of (ps, cb): k-slot (g, e) of the MFMA = pixel 16 ps + 8 g + e = image row 8 ps + 4 g + (e >> 1) + 16 (e & 1),
                    // column = channel 32 cb + (lane & 31).  ds_read_b64_tr_b16 (measured lane map, tools/micro/trprobe.hip): inside
                    // a 16-lane group, source lane 4 a + q reads four halfs, destination lane 4 q + c receives slot c of source lanes
                    // a = 0..3 as its elements 0..3.  So source lane (a, q) of group G (lane = 16 G + 4 a + q) reads channels
                    // 32 cb + 16 (G & 1) + 4 q .. + 4 of the image row of k-slot (G >> 1, e = a) — and of e = a + 4 (two rows further) for
                    // the second read; destination lane 16 G + 4 q + c = column 16 (G & 1) + 4 q + c gets its four consecutive k-slots.
                    const int tg = lane >> 4, ta = (lane >> 2) & 3, tq = lane & 3;
                    const _Float16* tp = dh + (4 * (tg >> 1) + (ta >> 1) + 16 * (ta & 1)) * LDK + 16 * (tg & 1) + 4 * tq;
                    half8 a[2][NB];
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            a[ps][nb] = lut[((wv[nb] >> (8 * ps + 4 * g)) & 0xFu) | (((wv[nb] >> (16 + 8 * ps + 4 * g)) & 0xFu) << 4)];
                    // steps (channel block j, 16-pixel half ps), j outer: every accumulator still sees ps 0 (hi, lo) then ps 1 (hi, lo);
                    // the fragments of step s + 2 are requested before the MFMAs of step s (3 fragment pairs live)
                    constexpr int NSTEP = 2 * CBW;
                    half8 fbh[3], fbl[3];
                    typedef __attribute__((address_space(3))) fs_short4 lds_s4;
                    auto ldf = [&](int slot, int st) {
                        const int cb = gw + 4 * (st >> 1);
                        if (cb < NCB) {
                            const _Float16* cp = tp + (8 * (st & 1)) * LDK + cb * 32;
                            const fs_short4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp));
                            const fs_short4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + 2 * LDK));
                            fbh[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                            if (!XH) {
                                const fs_short4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE));
                                const fs_short4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE + 2 * LDK));
                                fbl[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                            }
                        }
                    };
                    ldf(0, 0);
                    ldf(1, 1);
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        if (st + 2 < NSTEP) ldf((st + 2) % 3, st + 2);
                        const int j = st >> 1, ps = st & 1, cb = gw + 4 * j;
                        if (cb < NCB) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbh[st % 3], accg[j][nb], 0, 0, 0);
                                if (!XH) accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbl[st % 3], accg[j][nb], 0, 0, 0);
                            }
                        }
                    }
                } else {
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
            }
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(0);
            if constexpr ((V & 8) != 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_sched_barrier(0);
            }
            FS_STAMP(1);
#pragma unroll
            for (int f = 0; f < NF; ++f) commit((i + 1) % 3, f);
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(2);
#pragma unroll
            for (int f = 0; f < NF; ++f) issue(min(i + 2, tlast), f);
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(3);
            __syncthreads();
            FS_STAMP(4);
        }
#ifdef VKN_DEBUG
        if constexpr ((V & 8) != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 5; ++k) g_fs_prof[wave][k] = prof[k];
#endif
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

// ---------------------------------------------------------------------------------------------------------------------------
// k_fused_pp — the same roles in PING-PONG phases (round 3).  s_memtime stamps inside k_fused_dgs (V = 15, tools/perf_r03.py,
// profiles/r03_fused_phases.txt) showed, per 32-px tile of ~4700 cycles: both waves of a SIMD spend the same ~2400-3100 cycles in
// their MFMA phase (48 + 32 MFMAs = 2560 cycles of ONE shared matrix pipe) and then BOTH spend ~900-1300 cycles splitting / writing
// / requesting the next tile while the matrix pipe idles; and every wave waits 230-840 cycles for loads that were requested only
// ONE tile (32 KB per CU) ahead of a ~4000-cycle loaded-HBM latency.  Here:
//   * two phases per tile, one workgroup barrier each.  Phase A: decode waves run their 48 MFMAs + ballots on tile i WHILE the
//     gather waves split / write their share of tile i + 1, request tile i + 3 and fetch the operands (bit words -> table rows,
//     first transposed fragments) of tile i - 1.  Phase B: gather waves run their 32 MFMAs on tile i - 1 WHILE the decode waves do
//     their share of the loader work.  Each SIMD always has exactly one wave on the matrix pipe and its partner on VALU / LDS / VMEM.
//   * loads run TWO tiles ahead (two register sets per wave, 64 KB in flight per CU).
// Per-accumulator operation order is unchanged: results stay bit-identical to k_fused_dgs and to the unfused decode -> gather.
template <int NB, int C, int XH = 0, int PF = 0>
__global__ __launch_bounds__(FS_THREADS, 2) void k_fused_pp(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                             const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                             float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                             int N, int NPT, int n0, int P) {
    constexpr int KS = C / 16;
    constexpr int NF = (KS + 7) / 8;
    constexpr bool ALLF = (KS % 8) == 0;
    constexpr int NCB = C / 32;
    constexpr int CBW = (NCB + 3) / 4;
    constexpr int LDK = C + 8;
    constexpr int PLANE = FS_TILE * LDK;
    constexpr int IMG = 2 * PLANE;
    constexpr int V = PF ? 8 : 0;  // (FS_STAMP)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);         // [3 buffers][hi | lo][32 px][LDK]
    half8* lut = reinterpret_cast<half8*>(dimg + 3 * IMG);
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);   // [2 buffers][128 rows]
    float* kbs = reinterpret_cast<float*>(wbits + 256);         // [128]

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    (void)prof; (void)tprev;

    const int nsup = ((P >> 6) - gidx + G - 1) / G;
    const int T = 2 * nsup;  // 32-px tiles (always an even count): halves of the 64-px super-tiles s * G + gidx
    auto tile_p0 = [&](int t) { return (((t >> 1) * G + gidx) << 6) + ((t & 1) << 5); };

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
    const int lq = lane >> 4, lp = lane & 15;
    constexpr int XSH = XH ? 1 : 2;
    const int voff = (((lq << 2) * P + 2 * lp) << XSH);
    fu_u32x2 raw[2][NF][4];   // two tiles in flight
    auto issue = [&](int slot, int t, int f) {
        const int ks = wave + 8 * f;
        if (ALLF || ks < KS) {
            const int soff = ((ks << 4) * P + tile_p0(t)) << XSH;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (XH) raw[slot][f][e] = fu_u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, soff + ((e * P) << 1), 0), 0u};
                else raw[slot][f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
            }
        }
    };
    auto commit = [&](int slot, int buf, int f) {
        const int ks = wave + 8 * f;
        if (ALLF || ks < KS) {
            half4 h0, l0, h1, l1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u0 = raw[slot][f][e][0], u1 = raw[slot][f][e][1];
                _Float16 h, l;
                if (XH == 1) {
                    h0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 & 0xFFFFu));
                    h1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 >> 16));
                } else if (XH == 2) {
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
    // the loader step of one wave for iteration i (its "off" phase): tile i + 1 -> image (i + 1) % 3, request tile i + 3
    auto loader = [&](int slot, int i, int tlast) {
        FS_STAMP(0);
        if constexpr (PF != 0) {   // (profile build, C = 256: expose the wait for the OLDER of the two tiles in flight on its own)
            constexpr int vm = NF * 4;
            __builtin_amdgcn_s_waitcnt((vm & 0xF) | 0x0F70 | ((vm >> 4) << 14));
            __builtin_amdgcn_sched_barrier(0);
        }
        FS_STAMP(1);
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(slot, (i + 1) % 3, f);
        __builtin_amdgcn_sched_barrier(0);
        FS_STAMP(2);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(slot, min(i + 3, tlast), f);
        __builtin_amdgcn_sched_barrier(0);
        FS_STAMP(3);
    };

    const int tlast = max(T - 1, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) issue(0, 0, f);
#pragma unroll
    for (int f = 0; f < NF; ++f) issue(1, min(1, tlast), f);
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;

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
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) once: the prologue's loads (kernel rows, tiles 0 and 1) are complete — from
        __builtin_amdgcn_sched_barrier(0);   // here on the only vector loads in flight are the two tiles ahead
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(0, min(2, tlast), f);
        __syncthreads();
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i0 = 0; i0 < T; i0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = i0 + u;
                // ---------------------------------------------------- phase A: decode tile i
                if (has_dec) {
                    const _Float16* bp = dimg + (size_t)(i % 3) * IMG + li * LDK + (g << 3);
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = kbs[wave * 32 + vkn_cd_row(r, lane)];
                    half8 rbh[3], rbl[3];
                    auto ldb = [&](int slot, int ks) {
                        rbh[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4));
                        if (!XH) rbl[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                    };
                    ldb(0, 0);
                    if (KS > 1) ldb(1, 1);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) ldb((ks + 2) % 3, ks + 2);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbh[ks % 3], acc, 0, 0, 0);
                        if (!XH) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbl[ks % 3], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], rbh[ks % 3], acc, 0, 0, 0);
                    }
                    // [bias rows: 4 reads][k-steps 0, 1][k-step 2 | MFMAs 0][k-step 3 | MFMAs 1] ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (KS > 1 ? 2 : 1) * (XH ? 1 : 2), 0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, XH ? 1 : 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, XH ? 2 : 3, 0);
                    }
                    unsigned long long m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] >= thr);  // bit 32 g' + li' : row (r) + 4 g', image row li'
                    __builtin_amdgcn_sched_barrier(0);
                    int wd = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);
                        const int mlo = (int)(unsigned)m[r], mhi = (int)(unsigned)(m[r] >> 32);
                        FS_WRITELANE(wd, mlo, row);
                        FS_WRITELANE(wd, mhi, row + 4);
                    }
                    if (lane < 32) {
                        wbits[(i & 1) * 128 + wave * 32 + lane] = (unsigned)wd;
                        cnt_i += __popc((unsigned)wd);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                FS_STAMP(4);
                __syncthreads();
                FS_STAMP(5);
                // ---------------------------------------------------- phase B: this wave's share of the loader work
                loader(u ^ 1, i, tlast);
                __syncthreads();
                FS_STAMP(6);
            }
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
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
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) commit(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) issue(0, min(2, tlast), f);
        __syncthreads();
        // operands of one tile: table rows a[ps][nb] (bit words -> 8 halfs {0, 1}) and a 3-deep ring of transposed x fragments
        constexpr int NSTEP = 2 * CBW;   // (channel block j, 16-pixel half ps), j outer: every accumulator sees ps 0 (hi, lo), ps 1 (hi, lo)
        half8 a[2][NB], fbh[3], fbl[3];
        typedef __attribute__((address_space(3))) fs_short4 lds_s4;
        // ds_read_b64_tr_b16 lane map (tools/micro/trprobe.hip): see k_fused_dgs
        const int tg = lane >> 4, ta = (lane >> 2) & 3, tq = lane & 3;
        const int toff = (4 * (tg >> 1) + (ta >> 1) + 16 * (ta & 1)) * LDK + 16 * (tg & 1) + 4 * tq;
        auto ldf = [&](int slot, int st, const _Float16* dh) {
            const int cb = gw + 4 * (st >> 1);
            if (cb < NCB) {
                const _Float16* cp = dh + toff + (8 * (st & 1)) * LDK + cb * 32;
                const fs_short4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp));
                const fs_short4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + 2 * LDK));
                fbh[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                if (!XH) {
                    const fs_short4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE));
                    const fs_short4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE + 2 * LDK));
                    fbl[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
        };
        auto prep = [&](int t) {   // tile t: its bit words are complete (written in phase A of iteration t, two barriers ago)
            const _Float16* dh = dimg + (size_t)(t % 3) * IMG;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const unsigned wv = wbits[(t & 1) * 128 + nb * 32 + li];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
                    a[ps][nb] = lut[((wv >> (8 * ps + 4 * g)) & 0xFu) | (((wv >> (16 + 8 * ps + 4 * g)) & 0xFu) << 4)];
            }
            ldf(0, 0, dh);
            if (NSTEP > 1) ldf(1, 1, dh);
        };
        auto gather = [&](int t) {
            const _Float16* dh = dimg + (size_t)(t % 3) * IMG;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 2 < NSTEP) ldf((st + 2) % 3, st + 2, dh);
                const int j = st >> 1, ps = st & 1, cb = gw + 4 * j;
                if (cb < NCB) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbh[st % 3], accg[j][nb], 0, 0, 0);
                        if (!XH) accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbl[st % 3], accg[j][nb], 0, 0, 0);
                    }
                }
            }
        };
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i0 = 0; i0 < T; i0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = i0 + u;
                // ---------------------------------------------------- phase A: loader share, then the operands of tile i - 1
                loader(u ^ 1, i, tlast);
                if (i >= 1) prep(i - 1);
                __builtin_amdgcn_sched_barrier(0);
                FS_STAMP(4);
                __syncthreads();
                FS_STAMP(5);
                // ---------------------------------------------------- phase B: gather tile i - 1
                if (i >= 1) gather(i - 1);
                __builtin_amdgcn_sched_barrier(0);
                FS_STAMP(6);
                __syncthreads();
                FS_STAMP(7);
            }
        }
        if (T >= 1) {   // the last tile (the decode waves are done)
            prep(T - 1);
            gather(T - 1);
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
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

// ---------------------------------------------------------------------------------------------------------------------------
// k_fused_il — k_fused_dgs with every wave's loader work INTERLEAVED INTO its own MFMA stream (round 3, the shipped variant).
// What the phase stamps and ablations of k_fused_dgs / k_fused_pp / k_fused_pq established (profiles/r03_fused_phases_*.txt,
// profiles/r03_mfma_issue_rate.txt):
//   * one accumulator chain issues an MFMA every 52 cycles, two or more independent ones every 32; VALU instructions of the SAME wave
//     between its MFMAs are free (32.1 cycles per MFMA with 4 conversions in every gap);
//   * the partner wave of a SIMD gets almost no VALU issue while the other wave streams MFMAs (the same split + LDS-write code took
//     900 cycles beside a gather stream and 4100 beside a decode stream) — "one wave computes while its partner loads" does not
//     work here, ping-pong phases (k_fused_pp / k_fused_pq) were no faster than k_fused_dgs;
//   * in k_fused_dgs both waves of a SIMD ran their MFMAs at the same time (good: the gather MFMAs fill the 20-cycle gaps of the
//     decode wave's single dependent chain) and then BOTH did their loader work with the matrix pipe idle (bad: ~1300 of 4700
//     cycles per tile), and loads were requested one tile (32 KB per CU) ahead of a ~4000-cycle loaded-HBM latency.
// Here: same roles, same three images, one barrier per 32-px tile — but the split / LDS writes / load requests of a wave are
// scheduled INTO the gaps of its own MFMA stream (sched_group_barrier pipelines), loads run two tiles ahead (two register sets),
// and the ballot transposition is two asm blocks of 16 v_writelane.  Per-accumulator operation order is unchanged: bit-identical.
template <int NB, int C, int XH = 0, int PF = 0>
__global__ __launch_bounds__(FS_THREADS, 2) void k_fused_il(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                             const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                             float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                             int N, int NPT, int n0_in, int P, int NZ) {
    constexpr int KS = C / 16;
    // 16-channel fragments of a tile per loader wave: the decode waves (128 + 16 + 24 registers of operands) take NFD each, the gather
    // waves (128 + 32 + 24) NFG — at C = 256 3 + 1 (with two tiles in flight: 48 + 16 registers), else all of them go to the decode waves
    constexpr int NFG = KS / 16, NFD = KS / 4 - NFG, NF = NFD;
    static_assert(KS % 4 == 0 && 4 * (NFD + NFG) == KS, "C must be a multiple of 64");
    constexpr int NCB = C / 32;
    constexpr int CBW = (NCB + 3) / 4;
    // straight-line role loops: a branch per MFMA group costs accumulator copies at its join, so conditions that hold for every wave are
    // compile-time (all four n-blocks live / every gather wave owns CBW channel blocks), and the gather of "tile -1" in the first
    // iteration runs on a zeroed image and zero bit words instead of being branched around
    constexpr bool ALLDEC = (NB == 4), ALLCB = (NCB % 4) == 0;
    constexpr int LDK = C + 8;
    constexpr int PLANE = FS_TILE * LDK;
    constexpr int IMG = 2 * PLANE;
    constexpr int V = PF ? 8 : 0;  // (FS_STAMP)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);         // [3 buffers][hi | lo][32 px][LDK]
    half8* lut = reinterpret_cast<half8*>(dimg + 3 * IMG);
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);   // [2 buffers][128 rows]
    float* kbs = reinterpret_cast<float*>(wbits + 256);         // [128]

    // NZ > 1 (more than 128 kernel rows): NZ workgroups with ADJACENT block indices walk the same pixel tiles at the same time, each on
    // its own chunk of NB * 32 rows — the feature map is fetched from HBM once (the second reader hits the memory-side cache) instead
    // of once per chunk as with one launch per chunk
    const int b = blockIdx.y, gidx = blockIdx.x / NZ, G = gridDim.x / NZ;
    const int n0 = n0_in + (blockIdx.x - gidx * NZ) * (NB * 32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    (void)prof; (void)tprev;

    const int nsup = ((P >> 6) - gidx + G - 1) / G;
    const int T = 2 * nsup;  // 32-px tiles (always an even count): halves of the 64-px super-tiles s * G + gidx
    auto tile_p0 = [&](int t) { return (((t >> 1) * G + gidx) << 6) + ((t & 1) << 5); };

    for (int v = tid; v < 256; v += FS_THREADS) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
        lut[v] = h;
    }
    if (tid < 128) {
        const int n = n0 + tid;
        kbs[tid] = (kb && tid < NB * 32 && n < N) ? kb[(size_t)b * N + n] : 0.f;
        wbits[128 + tid] = 0u;                                   // bit words of "tile -1"
    }
    for (int v = tid; v < IMG / 8; v += FS_THREADS)              // image of "tile -1" (buffer 2): finite values for its 0 x x products
        reinterpret_cast<half8*>(dimg + 2 * IMG)[v] = half8{0, 0, 0, 0, 0, 0, 0, 0};

    const __amdgpu_buffer_rsrc_t xrs =
        XH ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P), 0,
                                               C * P * 2, 0x00020000)
           : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * C * P), 0, C * P * 4, 0x00020000);
    const int lq = lane >> 4, lp = lane & 15;
    constexpr int XSH = XH ? 1 : 2;
    const int voff = (((lq << 2) * P + 2 * lp) << XSH);
    fu_u32x2 raw[2][NF][4];   // two tiles in flight
    auto frag_ks = [&](int f) { return wave < 4 ? wave + 4 * f : 4 * NFD + (wave - 4) + 4 * f; };
    auto issue = [&](int slot, int t, int f) {
        const int ks = frag_ks(f);
        {
            const int soff = ((ks << 4) * P + tile_p0(t)) << XSH;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (XH) raw[slot][f][e] = fu_u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, soff + ((e * P) << 1), 0), 0u};
                else raw[slot][f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
            }
        }
    };
    auto commit = [&](int slot, int buf, int f) {
        const int ks = frag_ks(f);
        {
            half4 h0, l0, h1, l1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u0 = raw[slot][f][e][0], u1 = raw[slot][f][e][1];
                _Float16 h, l;
                if (XH == 1) {
                    h0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 & 0xFFFFu));
                    h1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 >> 16));
                } else if (XH == 2) {
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
    // the loader work of a tile in UNITS that the role loops place between their MFMAs: unit u of 2 * NF * 4 —
    //   u < NF * 4:  fragment f = u / 4, channel e = u % 4: split the channel's pixel pair into the fragment's four half4 rows (and
    //                write the four rows to the image after the fragment's last channel)
    //   else:        request the same (f, e) of tile `tnext` into the register set just consumed
    half4 ch0, cl0, ch1, cl1;
    auto loader_unit = [&](int nf, int slot, int buf, int tnext, int u) {
        const int NU = nf * 4;
        if (u < NU) {
            const int ff = u >> 2, e = u & 3, ks = frag_ks(ff);
            {
                const unsigned u0 = raw[slot][ff][e][0], u1 = raw[slot][ff][e][1];
                _Float16 h, l;
                if (XH == 1) {
                    ch0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 & 0xFFFFu));
                    ch1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 >> 16));
                } else if (XH == 2) {
                    ch0[e] = (_Float16)__uint_as_float(u0 << 16);
                    ch1[e] = (_Float16)__uint_as_float(u0 & 0xFFFF0000u);
                } else {
                    vkn_split_f16(__uint_as_float(u0), h, l);
                    ch0[e] = h;
                    cl0[e] = l;
                    vkn_split_f16(__uint_as_float(u1), h, l);
                    ch1[e] = h;
                    cl1[e] = l;
                }
                if (e == 3) {
                    _Float16* dh = dimg + (size_t)buf * IMG + lp * LDK + (ks << 4) + (lq << 2);
                    *reinterpret_cast<half4*>(dh) = ch0;
                    *reinterpret_cast<half4*>(dh + 16 * LDK) = ch1;
                    if (!XH) {
                        *reinterpret_cast<half4*>(dh + PLANE) = cl0;
                        *reinterpret_cast<half4*>(dh + 16 * LDK + PLANE) = cl1;
                    }
                }
            }
        } else if (u < 2 * NU) {
            const int v = u - NU, ff = v >> 2, e = v & 3, ks = frag_ks(ff);
            {
                const int soff = ((ks << 4) * P + tile_p0(tnext)) << XSH;
                if (XH) raw[slot][ff][e] = fu_u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, soff + ((e * P) << 1), 0), 0u};
                else raw[slot][ff][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
            }
        }
    };
    const int tlast = max(T - 1, 0);
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;

    if (wave < 4) {
        // =============================================================== decode role: n-block `wave`
        if constexpr (PF == 2) __builtin_amdgcn_s_setprio(2);   // (debug A/B: decode waves first at the issue arbiter)
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
        for (int f = 0; f < NFD; ++f) issue(0, 0, f);
#pragma unroll
        for (int f = 0; f < NFD; ++f) issue(1, min(1, tlast), f);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) once: the prologue's loads (kernel rows, tiles 0 and 1) are complete — from
        __builtin_amdgcn_sched_barrier(0);   // here on the only vector loads in flight are the two tiles ahead
#pragma unroll
        for (int f = 0; f < NFD; ++f) commit(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFD; ++f) issue(0, min(2, tlast), f);
        __syncthreads();
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i0 = 0; i0 < T; i0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = i0 + u;
                // the decode of tile i with this wave's loader units (tile i + 1 -> image (i + 1) % 3, request tile i + 3) in the gaps of
                // its MFMA chain: one unit behind every k-step, fenced so that the scheduler cannot pull them back together
                f32x16 acc;
                const _Float16* bp = dimg + (size_t)(i % 3) * IMG + li * LDK + (g << 3);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = kbs[wave * 32 + vkn_cd_row(r, lane)];
                {
                    half8 rbh[3], rbl[3];
                    auto ldb = [&](int slot, int ks) {
                        rbh[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4));
                        if (!XH) rbl[slot] = *reinterpret_cast<const half8*>(bp + (ks << 4) + PLANE);
                    };
                    ldb(0, 0);
                    if (KS > 1) ldb(1, 1);
                    constexpr int NUNIT = 2 * NFD * 4;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if (ks + 2 < KS) ldb((ks + 2) % 3, ks + 2);
                        if (ALLDEC || has_dec) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbh[ks % 3], acc, 0, 0, 0);
                            if (!XH) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbl[ks % 3], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], rbh[ks % 3], acc, 0, 0, 0);
                        }
                        // units spread over the k-steps (KS = 16: one per k-step; fewer k-steps: several per k-step)
#pragma unroll
                        for (int uu = (ks * NUNIT) / KS; uu < ((ks + 1) * NUNIT) / KS; ++uu) loader_unit(NFD, u ^ 1, (i + 1) % 3, min(i + 3, tlast), uu);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ALLDEC || has_dec) {
                    unsigned long long m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = __ballot(acc[r] >= thr);  // bit 32 g' + li' : row (r) + 4 g', image row li'
                    __builtin_amdgcn_sched_barrier(0);   // (v_writelane right behind the v_cmp that wrote its SGPR reads a stale value)
                    int wd = 0;
#define FS_WL2(r) (int)(unsigned)m[r], (int)(unsigned)(m[r] >> 32)
#define FS_WL8(wd, r0)                                                                                                              \
    asm volatile("v_writelane_b32 %0, %1, %17\n\tv_writelane_b32 %0, %2, %18\n\tv_writelane_b32 %0, %3, %19\n\tv_writelane_b32 %0, %4, %20\n\t"   \
                 "v_writelane_b32 %0, %5, %21\n\tv_writelane_b32 %0, %6, %22\n\tv_writelane_b32 %0, %7, %23\n\tv_writelane_b32 %0, %8, %24\n\t"   \
                 "v_writelane_b32 %0, %9, %25\n\tv_writelane_b32 %0, %10, %26\n\tv_writelane_b32 %0, %11, %27\n\tv_writelane_b32 %0, %12, %28\n\t" \
                 "v_writelane_b32 %0, %13, %29\n\tv_writelane_b32 %0, %14, %30\n\tv_writelane_b32 %0, %15, %31\n\tv_writelane_b32 %0, %16, %32"      \
                 : "+v"(wd)                                                                                                         \
                 : "s"((int)(unsigned)m[r0]), "s"((int)(unsigned)(m[r0] >> 32)), "s"((int)(unsigned)m[r0 + 1]),                     \
                   "s"((int)(unsigned)(m[r0 + 1] >> 32)), "s"((int)(unsigned)m[r0 + 2]), "s"((int)(unsigned)(m[r0 + 2] >> 32)),      \
                   "s"((int)(unsigned)m[r0 + 3]), "s"((int)(unsigned)(m[r0 + 3] >> 32)), "s"((int)(unsigned)m[r0 + 4]),              \
                   "s"((int)(unsigned)(m[r0 + 4] >> 32)), "s"((int)(unsigned)m[r0 + 5]), "s"((int)(unsigned)(m[r0 + 5] >> 32)),      \
                   "s"((int)(unsigned)m[r0 + 6]), "s"((int)(unsigned)(m[r0 + 6] >> 32)), "s"((int)(unsigned)m[r0 + 7]),              \
                   "s"((int)(unsigned)(m[r0 + 7] >> 32)), "n"(FS_ROW(r0)), "n"(FS_ROW(r0) + 4), "n"(FS_ROW(r0 + 1)),                 \
                   "n"(FS_ROW(r0 + 1) + 4), "n"(FS_ROW(r0 + 2)), "n"(FS_ROW(r0 + 2) + 4), "n"(FS_ROW(r0 + 3)), "n"(FS_ROW(r0 + 3) + 4), \
                   "n"(FS_ROW(r0 + 4)), "n"(FS_ROW(r0 + 4) + 4), "n"(FS_ROW(r0 + 5)), "n"(FS_ROW(r0 + 5) + 4), "n"(FS_ROW(r0 + 6)),   \
                   "n"(FS_ROW(r0 + 6) + 4), "n"(FS_ROW(r0 + 7)), "n"(FS_ROW(r0 + 7) + 4))
#define FS_ROW(r) (((r)&3) + 8 * ((r) >> 2))
                    FS_WL8(wd, 0);
                    FS_WL8(wd, 8);
                    if (lane < 32) {
                        wbits[(i & 1) * 128 + wave * 32 + lane] = (unsigned)wd;
                        cnt_i += __popc((unsigned)wd);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                FS_STAMP(4);
                __syncthreads();
                FS_STAMP(5);
            }
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
        if (has_dec && lane < 32 && n0 + wave * 32 + lane < NPT) cntp[((size_t)b * G + gidx) * NPT + n0 + wave * 32 + lane] = (float)cnt_i;
    } else {
        // =============================================================== gather role: channel blocks wave - 4 (+ 4)
        if constexpr (PF == 3) __builtin_amdgcn_s_setprio(2);   // (debug A/B: gather waves first)
        const int gw = wave - 4;
        f32x16 accg[CBW][NB];
#pragma unroll
        for (int j = 0; j < CBW; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[j][nb][r] = 0.f;
#pragma unroll
        for (int f = 0; f < NFG; ++f) issue(0, 0, f);
#pragma unroll
        for (int f = 0; f < NFG; ++f) issue(1, min(1, tlast), f);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFG; ++f) commit(0, 0, f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NFG; ++f) issue(0, min(2, tlast), f);
        __syncthreads();
        // operands of one tile: table rows a[ps][nb] (bit words -> 8 halfs {0, 1}) and a 3-deep ring of transposed x fragments
        constexpr int NSTEP = 2 * CBW;   // (channel block j, 16-pixel half ps), j outer: every accumulator sees ps 0 (hi, lo), ps 1 (hi, lo)
        half8 a[2][NB], fbh[3], fbl[3];
        typedef __attribute__((address_space(3))) fs_short4 lds_s4;
        // ds_read_b64_tr_b16 lane map (tools/micro/trprobe.hip): see k_fused_dgs
        const int tg = lane >> 4, ta = (lane >> 2) & 3, tq = lane & 3;
        const int toff = (4 * (tg >> 1) + (ta >> 1) + 16 * (ta & 1)) * LDK + 16 * (tg & 1) + 4 * tq;
        auto ldf = [&](int slot, int st, const _Float16* dh) {
            const int cb = gw + 4 * (st >> 1);
            if (ALLCB || cb < NCB) {
                const _Float16* cp = dh + toff + (8 * (st & 1)) * LDK + cb * 32;
                const fs_short4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp));
                const fs_short4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + 2 * LDK));
                fbh[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                if (!XH) {
                    const fs_short4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE));
                    const fs_short4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE + 2 * LDK));
                    fbl[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
        };
        auto prep = [&](int img, int wb) {   // image / bit-word buffer of the tile to gather (its bit words were written two barriers ago)
            const _Float16* dh = dimg + (size_t)img * IMG;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const unsigned wv = wbits[wb * 128 + nb * 32 + li];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
                    a[ps][nb] = lut[((wv >> (8 * ps + 4 * g)) & 0xFu) | (((wv >> (16 + 8 * ps + 4 * g)) & 0xFu) << 4)];
            }
            ldf(0, 0, dh);
            if (NSTEP > 1) ldf(1, 1, dh);
        };
        auto gather = [&](int t) {
            const _Float16* dh = dimg + (size_t)(t % 3) * IMG;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 2 < NSTEP) ldf((st + 2) % 3, st + 2, dh);
                const int j = st >> 1, ps = st & 1, cb = gw + 4 * j;
                if (ALLCB || cb < NCB) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbh[st % 3], accg[j][nb], 0, 0, 0);
                        if (!XH) accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbl[st % 3], accg[j][nb], 0, 0, 0);
                    }
                }
            }
        };
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int i0 = 0; i0 < T; i0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = i0 + u;
                // the gather of tile i - 1 with this wave's loader units in the gaps of its MFMA stream (two units behind every group of
                // four MFMAs), fenced against re-clustering
                prep((i + 2) % 3, (i + 1) & 1);   // tile i - 1 (i = 0: the zeroed image and bit words)
                {
                    const _Float16* dh = dimg + (size_t)((i + 2) % 3) * IMG;   // image of tile i - 1
                    constexpr int NUNIT = 2 * NFG * 4, NCH = NSTEP * (XH ? 1 : 2);
                    int ch = 0;
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        if (st + 2 < NSTEP) ldf((st + 2) % 3, st + 2, dh);
                        const int j = st >> 1, ps = st & 1, cb = gw + 4 * j;
#pragma unroll
                        for (int pl = 0; pl < (XH ? 1 : 2); ++pl) {
                            if (ALLCB || cb < NCB) {
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
                                    accg[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], pl ? fbl[st % 3] : fbh[st % 3], accg[j][nb], 0, 0, 0);
                            }
#pragma unroll
                            for (int uu = (ch * NUNIT) / NCH; uu < ((ch + 1) * NUNIT) / NCH; ++uu) loader_unit(NFG, u ^ 1, (i + 1) % 3, min(i + 3, tlast), uu);
                            ++ch;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                FS_STAMP(4);
                __syncthreads();
                FS_STAMP(5);
            }
        }
        if (T >= 1) {   // the last tile (the decode waves are done)
            prep((T - 1) % 3, (T - 1) & 1);
            gather(T - 1);
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
#pragma unroll
        for (int j = 0; j < CBW; ++j) {
            const int cb = gw + 4 * j;
            if (cb < NCB) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                        if (n < NPT) pp[(size_t)n * C + cb * 32 + li] = accg[j][nb][r];
                    }
            }
        }
    }
}

#ifdef VKN_DEBUG  // k_fused_w4 (one wave per SIMD, 64-px super-tiles; measured slower): tools/experiments/fused_w4.inc
#include "../../tools/experiments/fused_w4.inc"
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// k_fused_pq — ping-pong phases over 64-px PAIRS of strips (round 3, the shipped variant).  What the stamps of k_fused_pp added to
// the picture: a wave's MFMAs on ONE accumulator issue every ~48 cycles, not 32 (2335 cycles for the 48 MFMAs of a strip) — the
// dependent-accumulate latency of the 8-pass 32x32x16 MFMA; in k_fused_dgs the partner wave's MFMAs filled those gaps, in k_fused_pp
// nothing did.  A decode wave needs TWO independent chains, and the only way to get them without changing the summation order of
// any output element (the three stage hand-offs stay bit-identical) is two strips at once:
//   * a super-iteration handles the two 32-px strips of one 64-px super-tile; LDS holds two pair images (4 x 33.8 KB)
//   * phase A: decode waves run 96 MFMAs as two interleaved chains (dependent distance 64 cycles) + ballots of both strips,
//              gather waves do their share of the loader work (pair j + 1 -> the other pair image, request pair j + 2)
//   * phase B: gather waves fetch their operands and run 64 MFMAs on pair j (four independent accumulators between dependent ones),
//              decode waves do their share of the loader work
//   * a pair (64 KB per CU) is requested a whole super-iteration (~6 k cycles) ahead of its use: the ~4 k-cycle loaded-HBM latency
//     (measured from the stamps) is covered
template <int NB, int C, int XH = 0, int PF = 0>
__global__ __launch_bounds__(FS_THREADS, 2) void k_fused_pq(const float* __restrict__ x, const _Float16* __restrict__ kfh,
                                                             const _Float16* __restrict__ kfl, const float* __restrict__ kb,
                                                             float thr, float* __restrict__ part, float* __restrict__ cntp,
                                                             int N, int NPT, int n0, int P) {
    constexpr int KS = C / 16;
    constexpr int NF = (KS + 7) / 8;
    constexpr bool ALLF = (KS % 8) == 0;
    constexpr int NCB = C / 32;
    constexpr int CBW = (NCB + 3) / 4;
    constexpr int LDK = C + 8;
    constexpr int PLANE = FS_TILE * LDK;
    constexpr int IMG = 2 * PLANE;     // one strip: hi | lo
    constexpr int V = PF ? 8 : 0;      // (FS_STAMP)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* dimg = reinterpret_cast<_Float16*>(smem);         // [2 pair images][2 strips][hi | lo][32 px][LDK]
    half8* lut = reinterpret_cast<half8*>(dimg + 4 * IMG);
    unsigned* wbits = reinterpret_cast<unsigned*>(lut + 256);   // [2 strips][128 rows]
    float* kbs = reinterpret_cast<float*>(wbits + 256);         // [128]

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    (void)prof; (void)tprev;

    const int nsup = ((P >> 6) - gidx + G - 1) / G;   // 64-px super-tiles j * G + gidx of this workgroup
    const int jlast = max(nsup - 1, 0);

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
    const int lq = lane >> 4, lp = lane & 15;
    constexpr int XSH = XH ? 1 : 2;
    const int voff = (((lq << 2) * P + 2 * lp) << XSH);
    fu_u32x2 raw[2][NF][4];   // one pair in flight: [strip][fragment][channel]
    auto issue = [&](int j) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ks = wave + 8 * f;
                if (ALLF || ks < KS) {
                    const int soff = ((ks << 4) * P + ((j * G + gidx) << 6) + (s2 << 5)) << XSH;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (half storage: the two strips of a pair are the two halves of one 128-byte line per channel row)
                        if (XH) raw[s2][f][e] = fu_u32x2{__builtin_amdgcn_raw_buffer_load_b32(xrs, voff, soff + ((e * P) << 1), 0), 0u};
                        else raw[s2][f][e] = __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, soff + ((e * P) << 2), 3);
                    }
                }
            }
    };
    auto commit = [&](int pbuf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ks = wave + 8 * f;
                if (ALLF || ks < KS) {
                    half4 h0, l0, h1, l1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned u0 = raw[s2][f][e][0], u1 = raw[s2][f][e][1];
                        _Float16 h, l;
                        if (XH == 1) {
                            h0[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 & 0xFFFFu));
                            h1[e] = __builtin_bit_cast(_Float16, (unsigned short)(u0 >> 16));
                        } else if (XH == 2) {
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
                    _Float16* dh = dimg + (size_t)(2 * pbuf + s2) * IMG + lp * LDK + (ks << 4) + (lq << 2);
                    *reinterpret_cast<half4*>(dh) = h0;
                    *reinterpret_cast<half4*>(dh + 16 * LDK) = h1;
                    if (!XH) {
                        *reinterpret_cast<half4*>(dh + PLANE) = l0;
                        *reinterpret_cast<half4*>(dh + 16 * LDK + PLANE) = l1;
                    }
                }
            }
    };
    // the loader step of one wave in super-iteration j (its "off" phase): pair j + 1 -> pair image (j + 1) & 1, request pair j + 2
    auto loader = [&](int j) {
        // (profile builds PF = 2 / 3: the gather / decode waves skip their loader share — WRONG results, time attribution only)
        if constexpr (PF == 2) { if (wave >= 4) return; }
        if constexpr (PF == 3) { if (wave < 4) return; }
        FS_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the pair requested one super-iteration ago
        __builtin_amdgcn_sched_barrier(0);
        FS_STAMP(1);
        commit((j + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        FS_STAMP(2);
        issue(min(j + 2, jlast));
        __builtin_amdgcn_sched_barrier(0);
        FS_STAMP(3);
    };

    issue(0);
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;

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
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): kernel rows and pair 0 are here
        __builtin_amdgcn_sched_barrier(0);
        commit(0);
        __builtin_amdgcn_sched_barrier(0);
        issue(min(1, jlast));
        __syncthreads();
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int j = 0; j < nsup; ++j) {
            // -------------------------------------------------------- phase A: decode both strips of pair j, two independent chains
            if (has_dec) {
                const _Float16* bp = dimg + (size_t)(2 * (j & 1)) * IMG + li * LDK + (g << 3);
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float k0 = kbs[wave * 32 + vkn_cd_row(r, lane)];
                    acc0[r] = k0;
                    acc1[r] = k0;
                }
                half8 rbh[2][2], rbl[2][2];   // [ring slot][strip]
                auto ldb = [&](int slot, int ks) {
                    if constexpr (PF == 4) { if (ks > 1) return; }   // (profile build: no LDS reads after the first two k-steps — WRONG results)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        rbh[slot][s2] = *reinterpret_cast<const half8*>(bp + s2 * IMG + (ks << 4));
                        if (!XH) rbl[slot][s2] = *reinterpret_cast<const half8*>(bp + s2 * IMG + (ks << 4) + PLANE);
                    }
                };
                ldb(0, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) ldb((ks + 1) & 1, ks + 1);
                    const int sl = ks & 1;
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbh[sl][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbh[sl][1], acc1, 0, 0, 0);
                    if (!XH) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbl[sl][0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], rbl[sl][1], acc1, 0, 0, 0);
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], rbh[sl][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], rbh[sl][1], acc1, 0, 0, 0);
                }
                // [bias rows: 4 reads][k-step 0][k-step 1 | MFMAs 0][k-step 2 | MFMAs 1] ...
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, XH ? 2 : 4, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS && (PF != 4 || ks + 1 <= 1)) __builtin_amdgcn_sched_group_barrier(0x100, XH ? 2 : 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, XH ? 4 : 6, 0);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned long long m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = __ballot((s2 ? acc1[r] : acc0[r]) >= thr);  // bit 32 g' + li': row (r) + 4 g', image row li'
                    __builtin_amdgcn_sched_barrier(0);   // (v_writelane right behind the v_cmp that wrote its SGPR reads a stale value)
                    int wd = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);
                        const int mlo = (int)(unsigned)m[r], mhi = (int)(unsigned)(m[r] >> 32);
                        FS_WRITELANE(wd, mlo, row);
                        FS_WRITELANE(wd, mhi, row + 4);
                    }
                    if (lane < 32) {
                        wbits[s2 * 128 + wave * 32 + lane] = (unsigned)wd;
                        cnt_i += __popc((unsigned)wd);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(4);
            __syncthreads();
            FS_STAMP(5);
            // -------------------------------------------------------- phase B: this wave's share of the loader work
            loader(j);
            __syncthreads();
            FS_STAMP(6);
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
        if (has_dec && lane < 32) cntp[((size_t)b * G + gidx) * NPT + n0 + wave * 32 + lane] = (float)cnt_i;
    } else {
        // =============================================================== gather role: channel blocks wave - 4 (+ 4)
        const int gw = wave - 4;
        f32x16 accg[CBW][NB];
#pragma unroll
        for (int jj = 0; jj < CBW; ++jj)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[jj][nb][r] = 0.f;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        commit(0);
        __builtin_amdgcn_sched_barrier(0);
        issue(min(1, jlast));
        __syncthreads();
        constexpr int NSTEP = 2 * CBW;   // (channel block jj, 16-pixel half ps), jj outer: every accumulator sees ps 0 (hi, lo), ps 1 (hi, lo)
        typedef __attribute__((address_space(3))) fs_short4 lds_s4;
        // ds_read_b64_tr_b16 lane map (tools/micro/trprobe.hip): see k_fused_dgs
        const int tg = lane >> 4, ta = (lane >> 2) & 3, tq = lane & 3;
        const int toff = (4 * (tg >> 1) + (ta >> 1) + 16 * (ta & 1)) * LDK + 16 * (tg & 1) + 4 * tq;
#ifdef VKN_DEBUG
        if constexpr (PF != 0) tprev = __builtin_amdgcn_s_memtime();
#endif
        for (int j = 0; j < nsup; ++j) {
            // -------------------------------------------------------- phase A: loader share
            loader(j);
            FS_STAMP(4);
            __syncthreads();
            FS_STAMP(5);
            // -------------------------------------------------------- phase B: gather both strips of pair j
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const _Float16* dh = dimg + (size_t)(2 * (j & 1) + s2) * IMG;
                half8 a[2][NB], fbh[3], fbl[3];
                auto ldf = [&](int slot, int st) {
                    const int cb = gw + 4 * (st >> 1);
                    if (cb < NCB) {
                        const _Float16* cp = dh + toff + (8 * (st & 1)) * LDK + cb * 32;
                        const fs_short4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp));
                        const fs_short4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + 2 * LDK));
                        fbh[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        if (!XH) {
                            const fs_short4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE));
                            const fs_short4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(cp + PLANE + 2 * LDK));
                            fbl[slot] = __builtin_bit_cast(half8, (fs_short8)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                        }
                    }
                };
                ldf(0, 0);                 // (independent of the bit words: in flight while the table rows are looked up)
                if (NSTEP > 1) ldf(1, 1);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const unsigned wv = wbits[s2 * 128 + nb * 32 + li];
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps)
                        a[ps][nb] = lut[((wv >> (8 * ps + 4 * g)) & 0xFu) | (((wv >> (16 + 8 * ps + 4 * g)) & 0xFu) << 4)];
                }
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 2 < NSTEP) ldf((st + 2) % 3, st + 2);
                    const int jj = st >> 1, ps = st & 1, cb = gw + 4 * jj;
                    if (cb < NCB) {
                        // hi of every n-block, then lo of every n-block: four independent MFMAs between two on the same accumulator
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            accg[jj][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbh[st % 3], accg[jj][nb], 0, 0, 0);
                        if (!XH) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                accg[jj][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ps][nb], fbl[st % 3], accg[jj][nb], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            FS_STAMP(6);
            __syncthreads();
            FS_STAMP(7);
        }
#ifdef VKN_DEBUG
        if constexpr (PF != 0)
            if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
                for (int k = 0; k < 8; ++k) g_fs_prof[wave][k] = prof[k];
#endif
        // (the store offsets are derived from a laundered lane id: computed HERE, not hoisted above the loop and spilled across it)
        int lane2 = lane;
        asm volatile("" : "+v"(lane2));
#pragma unroll
        for (int jj = 0; jj < CBW; ++jj) {
            const int cb = gw + 4 * jj;
            if (cb < NCB) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + nb * 32 + vkn_cd_row(r, lane2);
                        pp[(size_t)n * C + cb * 32 + (lane2 & 31)] = accg[jj][nb][r];
                    }
            }
        }
    }
}

static size_t fusedq_lds_bytes(int C) { return (size_t)4 * 2 * FS_TILE * (C + 8) * sizeof(_Float16) + 256 * 16 + 256 * 4 + 128 * 4; }

#ifdef VKN_DEBUG
static size_t fusedw4_lds_bytes(int C) { return (size_t)4 * 2 * FS_TILE * (C + 8) * sizeof(_Float16) + 256 * 16 + 256 * 4 + 128 * 4; }
#endif
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
    const int variant = vkn_dbg_env("VKN_FUSED", 10);  // debug build A/B: 0 = k_fused_dg, 1 = k_fused_dg8, 2 = k_fused_dgs, 3 = k_fused_pp (4: profile), 5 = k_fused_pq (6: profile), 10 = k_fused_il (shipped; 11: profile)
    const bool eight = variant == 1;
    const size_t lds = variant >= 2 ? fuseds_lds_bytes(C) : (eight ? fused8_lds_bytes(C) : fused_lds_bytes(C));
#else
    const size_t lds = fuseds_lds_bytes(C);
#endif
#ifdef VKN_DEBUG
    const bool one_pass = (variant >= 10 && variant <= 18) && !vkn_dbg_env("VKN_FUSED_CHUNK_LOOP", 0);
#else
    const bool one_pass = true;
#endif
    // more than 128 rows: NZ equal chunks handled by NZ adjacent workgroups of ONE launch (k_fused_il), see the kernel
    const int NZ = one_pass ? (NPT + 127) / 128 : 1;
    const int nb_one = (NPT / 32 + NZ - 1) / NZ;
    for (int n0 = 0; n0 < (one_pass ? 1 : NPT); n0 += 128) {
        const int nb = one_pass ? nb_one : ((NPT - n0 >= 128) ? 4 : (NPT - n0) / 32);
        dim3 grid(G * NZ, B, 1);
#define FU_LAUNCH_XV(NBV, CV, XHV, VV)                                                                                         \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_dgs<NBV, CV, XHV, VV>));                                                                   \
        hipLaunchKernelGGL((k_fused_dgs<NBV, CV, XHV, VV>), grid, dim3(FS_THREADS), lds, stream, x, kfh, kfl, kb, thr, part,   \
                           cntp, N, NPT, n0, P);                                                                               \
    } while (0)
#define FU_LAUNCH_PQ(NBV, CV, XHV, PFV)                                                                                        \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_pq<NBV, CV, XHV, PFV>));                                                                   \
        hipLaunchKernelGGL((k_fused_pq<NBV, CV, XHV, PFV>), grid, dim3(FS_THREADS), fusedq_lds_bytes(C), stream, x, kfh, kfl,  \
                           kb, thr, part, cntp, N, NPT, n0, P);                                                                \
    } while (0)
#define FU_LAUNCH_IL(NBV, CV, XHV, PFV)                                                                                        \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_il<NBV, CV, XHV, PFV>));                                                                   \
        hipLaunchKernelGGL((k_fused_il<NBV, CV, XHV, PFV>), grid, dim3(FS_THREADS), lds, stream, x, kfh, kfl, kb, thr, part,   \
                           cntp, N, NPT, n0, P, NZ);                                                                           \
    } while (0)
#ifdef VKN_DEBUG
#define FU_LAUNCH_W4(NBV, CV, XHV, PFV)                                                                                        \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_w4<NBV, CV, XHV, PFV>));                                                                   \
        hipLaunchKernelGGL((k_fused_w4<NBV, CV, XHV, PFV>), grid, dim3(W4_THREADS), fusedw4_lds_bytes(C), stream, x, kfh, kfl,  \
                           kb, thr, part, cntp, N, NPT, n0, P, NZ);                                                            \
    } while (0)
#endif
#define FU_LAUNCH_PP(NBV, CV, XHV, PFV)                                                                                        \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_pp<NBV, CV, XHV, PFV>));                                                                   \
        hipLaunchKernelGGL((k_fused_pp<NBV, CV, XHV, PFV>), grid, dim3(FS_THREADS), lds, stream, x, kfh, kfl, kb, thr, part,   \
                           cntp, N, NPT, n0, P);                                                                               \
    } while (0)
#ifdef VKN_DEBUG  // A/B (VKN_FUSED: 3 = k_fused_pp (shipped), 4 = its profile build, 2 = k_fused_dgs with VKN_FUSED_V = 0 .. 7 | 15)
#define FU_LAUNCH_X(NBV, CV, XHV)                                                          \
    do {                                                                                   \
        const int vv = vkn_dbg_env("VKN_FUSED_V", FS_V_DEFAULT);                           \
        const bool cfg2 = NBV == 4 && CV == 256 && XHV == 0;                               \
        if (variant == 4 && cfg2) FU_LAUNCH_PP(4, 256, 0, 1);                              \
        else if (variant == 6 && cfg2) FU_LAUNCH_PQ(4, 256, 0, 1);                         \
        else if (variant == 7 && cfg2) FU_LAUNCH_PQ(4, 256, 0, 2);                         \
        else if (variant == 8 && cfg2) FU_LAUNCH_PQ(4, 256, 0, 3);                         \
        else if (variant == 9 && cfg2) FU_LAUNCH_PQ(4, 256, 0, 4);                         \
        else if (variant == 3) FU_LAUNCH_PP(NBV, CV, XHV, 0);                              \
        else if (variant == 10) FU_LAUNCH_IL(NBV, CV, XHV, 0);                             \
        else if (variant == 12) FU_LAUNCH_W4(NBV, CV, XHV, 0);                             \
        else if (variant == 13 && cfg2) FU_LAUNCH_W4(4, 256, 0, 1);                        \
        else if (variant == 14 && cfg2) FU_LAUNCH_W4(4, 256, 0, 2);                        \
        else if (variant == 15 && cfg2) FU_LAUNCH_W4(4, 256, 0, 3);                        \
        else if (variant == 16 && cfg2) FU_LAUNCH_W4(4, 256, 0, 4);                        \
        else if (variant == 11 && cfg2) FU_LAUNCH_IL(4, 256, 0, 1);                        \
        else if (variant == 17 && cfg2) FU_LAUNCH_IL(4, 256, 0, 2);                        \
        else if (variant == 18 && cfg2) FU_LAUNCH_IL(4, 256, 0, 3);                        \
        else if (variant == 5) FU_LAUNCH_PQ(NBV, CV, XHV, 0);                              \
        else if (cfg2 && vv != FS_V_DEFAULT) {                                             \
            switch (vv) {                                                                  \
                case 0: FU_LAUNCH_XV(4, 256, 0, 0); break;                                 \
                case 1: FU_LAUNCH_XV(4, 256, 0, 1); break;                                 \
                case 2: FU_LAUNCH_XV(4, 256, 0, 2); break;                                 \
                case 3: FU_LAUNCH_XV(4, 256, 0, 3); break;                                 \
                case 4: FU_LAUNCH_XV(4, 256, 0, 4); break;                                 \
                case 5: FU_LAUNCH_XV(4, 256, 0, 5); break;                                 \
                case 15: FU_LAUNCH_XV(4, 256, 0, 15); break;                               \
                default: FU_LAUNCH_XV(4, 256, 0, 6); break;                                \
            }                                                                              \
        } else FU_LAUNCH_XV(NBV, CV, XHV, FS_V_DEFAULT);                                   \
    } while (0)
#else
#define FU_LAUNCH_X(NBV, CV, XHV) FU_LAUNCH_IL(NBV, CV, XHV, 0)
#endif
#define FU_LAUNCH_S(NBV, CV)                        \
    do {                                            \
        if (xdt == 1) FU_LAUNCH_X(NBV, CV, 1);      \
        else if (xdt == 2) FU_LAUNCH_X(NBV, CV, 2); \
        else FU_LAUNCH_X(NBV, CV, 0);               \
    } while (0)
#ifdef VKN_DEBUG
#define FU_LAUNCH(NBV, CV)                                                                                                     \
    do {                                                                                                                       \
        if (variant >= 2 || xdt != 0) FU_LAUNCH_S(NBV, CV);                                                                    \
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
#undef FU_LAUNCH_XV
#undef FU_LAUNCH_PP
#undef FU_LAUNCH_IL
#ifdef VKN_DEBUG
#undef FU_LAUNCH_W4
#endif
#undef FU_LAUNCH_PQ
        VKN_CHECK_LAUNCH();
    }
    return vkn_launch_gather_reduce(part, cntp, xraw, cnt, B, N, C, G, stream);  // the unfused path's fixed-order second pass
}

#ifdef VKN_DEBUG
// debug library only: the s_memtime phase sums of the last V = 15 launch (workgroup (0, 0), [wave][phase]), synchronous
extern "C" int vkn_dbg_fused_prof(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_fs_prof), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif
