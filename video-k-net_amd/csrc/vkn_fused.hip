// vkn_fused.hip — mask decode of stage s FUSED with the mask gather of stage s + 1: ONE pass over x per stage boundary.
//
//   z[n][p]      = kb[n] + sum_c K[n][c] x[c][p]                    (stage s decode,  knet/det/kernel_update_head.py:247-260)
//   bit[n][p]    = z[n][p] >= thr_logit                             (stage s+1 binarise,                        :190-192)
//   xraw[n][c]   = sum_p bit[n][p] x[c][p],  cnt[n] = sum_p bit     (stage s+1 gather `einsum('bnhw,bchw->bnc')`, :195)
//
// The logits of an intermediate stage are consumed by nothing but this threshold, so neither they nor
// their bit words ever reach HBM, and x is streamed once where k_decode_mfma<bits> + k_gather_bits_w streamed it twice.
//
// ONE kernel ships: k_fused_il (below) — 8 waves per workgroup, role-specialised (4 decode + 4 gather, one of each per SIMD), 32-px
// tiles through three LDS images, every wave's loader work interleaved into its own MFMA stream.  FlashAttention-shaped
// (S = K x -> P = bit(S) -> O += P x^T per tile) and bit-identical to the unfused k_decode_mfma<bits> -> k_gather_bits_w path.
// Its predecessors and the rejected designs (k_fused_dg / dg8 / dgs / pp / pq / w4) are built into the DEBUG library only, from
// tools/experiments/*.inc; what each one measured is in DESIGN.md §6.
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
// Roles (k_fused_il, and k_fused_dgs before it): the convert -> decode -> gather phases of a tile run CONCURRENTLY on different
// waves of each SIMD:
//   waves 0-3 (one per SIMD): decode role, n-block = wave; kernel rows (hi + lo) stationary in registers; tile i
//   waves 4-7 (one per SIMD): gather role, channel blocks wave - 4 and wave; tile i - 1 (its bit words are ready)
//   all waves: load / split / write 1/8 of tile i + 1 into the third image buffer, request tile i + 2 / i + 3
// Tiles are 32 px (one MFMA strip): three image buffers fit (3 x 33.8 KB), ONE workgroup barrier per tile.  The 32-px tiles are
// walked in the order of the 64-px super-tiles' halves and the MFMA k index maps to the same pixels, so partials and results stay
// bit-identical to the unfused path.
#define FS_THREADS 512
#define FS_TILE 32

// XH (x storage): 0 = fp32; 1 = fp16, 2 = bf16 (converted to f16): the loader's pixel pair is one dword, only the hi plane of the
// tile image is written / read and every MFMA against x_lo disappears (decode 3 -> 2, gather 2 -> 1 per operand pair).  On
// x' = float(half(x)) the fp32 kernel returns the same bits.
// V (variant bits of k_fused_dgs, debug library; k_fused_il has all of 1 | 2 | 4 built in — all bit-identical):
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

#ifdef VKN_DEBUG  // k_fused_dgs (round 2), k_fused_pp / k_fused_pq (ping-pong phases; no faster): tools/experiments/fused_older.inc
#include "../../tools/experiments/fused_older.inc"
#endif

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
        if (!XH && !VKN_ABL_IS(PF, 4) && u < NU) {   // (debug arm PF == 4: the round-3 split, one channel's pixel pair per unit)
            // fp32 x: unit j of a fragment = pixel j >> 1, channel pair j & 1 — the two channels of ONE pixel are split together
            // (vkn_split_f16x2: four VALU operations per pair, and the packed results are already the dwords of the pixel's half4)
            const int ff = u >> 2, j = u & 3, px = j >> 1, ep = j & 1, ks = frag_ks(ff);
            vkn_half2 h2, l2;
            vkn_split_f16x2(__uint_as_float(raw[slot][ff][2 * ep][px]), __uint_as_float(raw[slot][ff][2 * ep + 1][px]), h2, l2);
            if (px == 0) {
                ch0[2 * ep] = h2[0]; ch0[2 * ep + 1] = h2[1];
                cl0[2 * ep] = l2[0]; cl0[2 * ep + 1] = l2[1];
            } else {
                ch1[2 * ep] = h2[0]; ch1[2 * ep + 1] = h2[1];
                cl1[2 * ep] = l2[0]; cl1[2 * ep + 1] = l2[1];
            }
            if (j == 3) {
                _Float16* dh = dimg + (size_t)buf * IMG + lp * LDK + (ks << 4) + (lq << 2);
                *reinterpret_cast<half4*>(dh) = ch0;
                *reinterpret_cast<half4*>(dh + 16 * LDK) = ch1;
                *reinterpret_cast<half4*>(dh + PLANE) = cl0;
                *reinterpret_cast<half4*>(dh + 16 * LDK + PLANE) = cl1;
            }
        } else if (u < NU) {
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
                                   hipStream_t stream, int xdt, int* status, const void* touch, size_t touch_bytes) {
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
    const bool one_pass = (variant >= 10 && variant <= 19) && !vkn_dbg_env("VKN_FUSED_CHUNK_LOOP", 0);
#else
    const bool one_pass = true;
#endif
    // more than 128 rows: NZ equal chunks handled by NZ adjacent workgroups of ONE launch (k_fused_il), see the kernel
    const int NZ = one_pass ? (NPT + 127) / 128 : 1;
    const int nb_one = (NPT / 32 + NZ - 1) / NZ;
    for (int n0 = 0; n0 < (one_pass ? 1 : NPT); n0 += 128) {
        const int nb = one_pass ? nb_one : ((NPT - n0 >= 128) ? 4 : (NPT - n0) / 32);
        dim3 grid(G * NZ, B, 1);
#ifdef VKN_DEBUG
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
#endif
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
#ifdef VKN_DEBUG
#define FU_LAUNCH_PP(NBV, CV, XHV, PFV)                                                                                        \
    do {                                                                                                                       \
        VKN_ALLOW_FULL_LDS((k_fused_pp<NBV, CV, XHV, PFV>));                                                                   \
        hipLaunchKernelGGL((k_fused_pp<NBV, CV, XHV, PFV>), grid, dim3(FS_THREADS), lds, stream, x, kfh, kfl, kb, thr, part,   \
                           cntp, N, NPT, n0, P);                                                                               \
    } while (0)
#endif
#ifdef VKN_DEBUG  // A/B (VKN_FUSED: 10 = k_fused_il (shipped), 11 = its profile build, 3 / 5 = k_fused_pp / k_fused_pq, 2 = k_fused_dgs with VKN_FUSED_V = 0 .. 7 | 15)
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
        else if (variant == 19 && cfg2) FU_LAUNCH_IL(4, 256, 0, 4);                        \
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
#undef FU_LAUNCH_IL
#ifdef VKN_DEBUG
#undef FU_LAUNCH_XV
#undef FU_LAUNCH_PP
#undef FU_LAUNCH_PQ
#undef FU_LAUNCH_W4
#endif
        VKN_CHECK_LAUNCH();
    }
    return vkn_launch_gather_reduce(part, cntp, xraw, cnt, B, N, C, G, stream, status, touch, touch_bytes);  // the unfused path's fixed-order second pass
}

#ifdef VKN_DEBUG
// debug library only: the s_memtime phase sums of the last V = 15 launch (workgroup (0, 0), [wave][phase]), synchronous
extern "C" int vkn_dbg_fused_prof(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_fs_prof), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif
