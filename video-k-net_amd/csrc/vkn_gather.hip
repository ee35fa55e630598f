// vkn_gather.hip — mask gather ("group feature assembling"):
//     xraw[b][n][c] = sum_p bit(mask_logits[b][n][p]) * x[b][c][p],   cnt[b][n] = sum_p bit(...)
// with bit(z) = (z >= thr_logit)  <=>  the reference's  (sigmoid(z) > hard_mask_thr).float()
// (knet/det/kernel_update_head.py:190-195 `einsum('bnhw,bchw->bnc')`; thr_logit is the smallest fp32 z for which the
// fp32 sigmoid exceeds the threshold — 8.940697e-08 for 0.5, found by the host by bisection, see ops.py).
// The per-stage `feat_transform` 1x1 conv is folded AFTER the gather by the update kernels:
//     x_feat = xraw . W_ft^T + cnt (x) b_ft          (SURVEY.md §7 "Folding feat_transform")
//
// MI355X design: HBM-bound stream of x and of the mask logits, each read once.
//   * contraction index = pixel (contiguous in memory), so fragments need a [row][pixel] transpose: tiles of 32 pixels are
//     loaded with coalesced dwordx4 (8 lanes per 128-B row segment), split to f16 hi/lo (x) or binarised to f16 {0,1}
//     (mask) in registers and written to a padded LDS image (row stride 80 B -> conflict-free ds_read_b128);
//   * wave w owns channel block w (32 channels) x all n-blocks: 2 MFMA (m*x_hi + m*x_lo, exact products) per
//     (n-block, 16 px); waves 0..NB-1 also run one MFMA against an all-ones fragment to count pixels per kernel;
//   * LDS double-buffered, next tile's global loads in flight during the MFMAs, one barrier per tile;
//   * each workgroup walks a contiguous pixel range of one frame and writes one [NPT][C] fp32 partial; partials are
//     summed in fixed order by k_gather_reduce (deterministic, no atomics).
#include "vkn_common.h"
#include "vkn_launch.h"

#define GA_THREADS 512
#define GA_WAVES 8
#define GA_PT 32   // pixels per tile
#define GA_LDR 40  // halfs per LDS row (32 + 8 pad)

// BITS == 2: `masks` is a REAL-valued left operand a[b][n][p] (soft gather weights, soft ground-truth masks): out = sum_p a x,
// `cnt` = sum_p a; both operands f16 hi / lo split, a_hi x_hi + a_hi x_lo + a_lo x_hi (|a|, |x| < 65504; absolute resolution of a: 6e-8).
// BITS == 1: `masks` are the bit words [B][P/64][2][NPT] (even / odd pixels of each 64-px tile) written by the decode kernel's
// bit-packed epilogue (fused head, stages > 0) instead of logits; fragments of the binary operand come from a 256-entry
// (even nibble, odd nibble) -> half8 table in LDS.
// XH (x storage, BITS 0 / 1 only): 0 = fp32; 1 = fp16, 2 = bf16 (converted to f16) — x is its own high half: a 16-byte chunk holds
// 8 pixels (two chunks per thread and tile instead of four), only the hi image is written and the MFMA against x_lo disappears.
// Needs P % 64 == 0 (whole, 16-byte aligned tiles).  On x' = float(half(x)) the fp32 kernel returns the same bits.
template <int NB, int BITS, int XH = 0>
__global__ __launch_bounds__(GA_THREADS, 2) void k_gather_mfma(const float* __restrict__ x,
                                                               const float* __restrict__ masks, float thr,
                                                               float* __restrict__ part, float* __restrict__ cntp, int N,
                                                               int NPT, int n0, int C, int P, int px_per_wg,
                                                               long long mask_fs, int ileave, int x_alias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // x_alias >= 0 (BITS == 2 only; the assignment's cost contraction): channels c >= x_alias of the feature operand are not stored —
    // they are max(x[c - x_alias], thr), formed on the way into LDS (`thr` is otherwise unused by this mode).  The second activation
    // of MaskCost / DiceCost is the first one clamped higher: one plane written and streamed instead of two.
    // per buffer: xh [C][40], xl [C][40], mk [NB*32][40] (+ ml [NB*32][40]: low half of a REAL mask operand, BITS == 2)
    constexpr bool REAL = (BITS == 2 || BITS == 3);  // 3: real operand = bit(z) * sigmoid(z), activated on the fly
    static_assert(!(XH && REAL), "half-storage x: binary operands only");
    constexpr int XCH = XH ? 2 : 4;  // 16-byte x chunks per thread and tile (C = 256)
    const int rows_buf = 2 * C + (REAL ? 2 : 1) * NB * 32;
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    const int g = lane >> 5, li = lane & 31;

    // Pixel ranges.  ileave == 0: workgroup g walks the contiguous range [g, g+1) * px_per_wg.  ileave == 1 (P % 64 == 0): it walks
    // the 64-px super-tiles j * G + g, j = 0, 1, ... — at any moment the G workgroups of a frame read ADJACENT 256-byte pieces of
    // every channel row (DRAM page locality across CUs) instead of G pieces 4 KB apart.  Both gather kernels visit the 32-px
    // tiles of a workgroup in the same order, so their partial sums are bit-identical.
    const int p_begin = ileave ? 0 : gidx * px_per_wg;
    const int p_end = ileave ? 0 : min(P, p_begin + px_per_wg);
    const int nsup_i = ileave ? (((P >> 6) - gidx + G - 1) / G) : 0;
    const int ntiles = ileave ? 2 * nsup_i : ((p_end > p_begin) ? (p_end - p_begin + GA_PT - 1) / GA_PT : 0);
    auto tile_p0 = [&](int t) { return ileave ? ((((t >> 1) * G + gidx) << 6) + ((t & 1) << 5)) : p_begin + t * GA_PT; };

    const float* xb = x + (size_t)b * C * P;
    const unsigned short* xb16 = reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P;  // XH
    const float* mb = masks + (size_t)b * mask_fs;  // mask_fs = rows per frame of the logits tensor * P (>= N * P)
    const unsigned* wb = reinterpret_cast<const unsigned*>(masks) + (size_t)b * (P >> 5) * NPT + n0;  // BITS: words of this frame / n-chunk
    const bool vec_ok = ((P & 3) == 0);
    // BITS: (even nibble | odd nibble << 4) -> 8 halfs {0,1}: pixel e of the 8-px group = bit e/2 of the even (e even) or odd
    // (e odd) nibble; behind the two tile buffers
    half8* lut = reinterpret_cast<half8*>(lds + (size_t)2 * rows_buf * GA_LDR);
    if (BITS == 1) {
        for (int v = tid; v < 256; v += GA_THREADS) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
            lut[v] = h;
        }
    }

    const int nxch = XH ? C * 4 : C * 8;  // 16-B chunks in an x tile
    const int nmch = NB * 32 * 8;  // 16-B chunks in a mask tile
    f32x4 xrA[4], xrB[4];  // two register sets: tile t+2 is being loaded while tile t+1 waits to be committed
    f32x4 mrA[2], mrB[2];

    const float off_v = (BITS == 2) ? 0.f : -INFINITY;  // "off" for padded rows / pixels
    const f32x4 neg_inf = {off_v, off_v, off_v, off_v};

    // Full tiles (all 32 px in range, rows 16-B aligned): branch-free, ALWAYS 4 + 2 dwordx4 loads per thread on clamped
    // addresses (lanes past the tile are masked in commit()), nothing consumed here -> exact vmcnt counting in the pipeline.
    auto issue = [&](int t, f32x4 (&xr)[4], f32x4 (&mr)[2]) {
        const int p0 = tile_p0(t);
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int idc = min(tid + i * GA_THREADS, nxch - 1);
            // non-temporal: x is streamed once per launch (same +9 % as in the decode kernel)
            if (XH)  // plain load: the 32-px tile is half a 128-byte line, the next tile of this workgroup takes the other half
                xr[i] = *reinterpret_cast<const f32x4*>(xb16 + (size_t)(idc >> 2) * P + p0 + ((idc & 3) << 3));
            else if (BITS == 2 && x_alias >= 0) {   // (plain loads: every stored row is read twice — once as itself, once as its alias)
                const int ch = idc >> 3;
                xr[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)(ch >= x_alias ? ch - x_alias : ch) * P + p0 + ((idc & 7) << 2));
            } else
                xr[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + (size_t)(idc >> 3) * P + p0 + ((idc & 7) << 2)));
        }
        if (BITS == 1) {  // even- and odd-pixel word of this row for the 64-px tile holding p0 (clamped: every thread loads)
            const unsigned* wp = wb + (size_t)((p0 >> 6) << 1) * NPT + min(tid, NB * 32 - 1);
            const unsigned we = wp[0], wo = wp[NPT];
            mr[0][0] = __uint_as_float(we);
            mr[0][1] = __uint_as_float(wo);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idc = min(tid + i * GA_THREADS, nmch - 1);
                const int n = min(n0 + (idc >> 3), N - 1);
                mr[i] = *reinterpret_cast<const f32x4*>(mb + (size_t)n * P + p0 + ((idc & 7) << 2));
            }
        }
    };

    // Ragged tile (frame tail) or P % 4 != 0: guarded scalar loads, not pipelined.
    auto issue_slow = [&](int t, f32x4 (&xr)[4], f32x4 (&mr)[2]) {
        const int p0 = tile_p0(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * GA_THREADS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < nxch) {
                const int p = p0 + ((idx & 7) << 2);
                const int ch = idx >> 3;
                const float* src = xb + (size_t)((BITS == 2 && x_alias >= 0 && ch >= x_alias) ? ch - x_alias : ch) * P + p;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (p + k < p_end) v[k] = src[k];
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * GA_THREADS;
            f32x4 v = neg_inf;
            if (idx < nmch) {
                const int n = n0 + (idx >> 3), p = p0 + ((idx & 7) << 2);
                if (n < N) {
                    const float* src = mb + (size_t)n * P + p;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (p + k < p_end) v[k] = src[k];
                }
            }
            mr[i] = v;
        }
    };

    auto commit = [&](int buf, const f32x4 (&xr)[4], const f32x4 (&mr)[2]) {
        _Float16* xh = lds + (size_t)buf * rows_buf * GA_LDR;
        _Float16* xl = xh + C * GA_LDR;
        _Float16* mk = xl + C * GA_LDR;
        if (XH) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int idx = tid + i * GA_THREADS;
                if (idx < nxch) {
                    f32x4 v = xr[i];
                    if (XH == 2) {  // 8 bf16 -> 8 f16 (through fp32: exact for normal f16 range)
                        half8 h;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned u = __float_as_uint(xr[i][k]);
                            h[2 * k] = (_Float16)__uint_as_float(u << 16);
                            h[2 * k + 1] = (_Float16)__uint_as_float(u & 0xFFFF0000u);
                        }
                        v = __builtin_bit_cast(f32x4, h);
                    }
                    *reinterpret_cast<f32x4*>(xh + (idx >> 2) * GA_LDR + ((idx & 3) << 3)) = v;
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * GA_THREADS;
            if (idx < nxch) {
                half4 h, l;
                const bool floor_ch = (BITS == 2) && x_alias >= 0 && (idx >> 3) >= x_alias;   // an aliased channel: max(stored value, thr)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    _Float16 hh, ll;
                    float xv = xr[i][k];
                    if (BITS == 2) xv = floor_ch ? fmaxf(xv, thr) : xv;
                    vkn_split_f16(xv, hh, ll);
                    h[k] = hh;
                    l[k] = ll;
                }
                const int off = (idx >> 3) * GA_LDR + ((idx & 7) << 2);
                *reinterpret_cast<half4*>(xh + off) = h;
                *reinterpret_cast<half4*>(xl + off) = l;
            }
        }
        if (BITS == 1) {
            const float w0 = mr[0][0], w1 = mr[0][1];
            if (tid < NB * 32) {
                reinterpret_cast<unsigned*>(mk)[tid] = __float_as_uint(w0);
                reinterpret_cast<unsigned*>(mk)[NB * 32 + tid] = __float_as_uint(w1);
            }
        } else
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * GA_THREADS;
            if (idx < nmch) {
                const bool row_ok = (n0 + (idx >> 3)) < N;
                half4 m, ml;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (REAL) {  // real-valued left operand: f16 hi / lo split like x
                        _Float16 hh, ll;
                        float av = row_ok ? mr[i][k] : 0.f;
                        if (BITS == 3)  // soft gather weights (sigmoid(z) > thr) * sigmoid(z)   knet/det/kernel_head.py:243-249
                            av = (row_ok && mr[i][k] >= thr) ? 1.0f / (1.0f + expf(-mr[i][k])) : 0.f;
                        vkn_split_f16(av, hh, ll);
                        m[k] = hh;
                        ml[k] = ll;
                    } else {
                        m[k] = (row_ok && mr[i][k] >= thr) ? (_Float16)1.f : (_Float16)0.f;
                    }
                }
                *reinterpret_cast<half4*>(mk + (idx >> 3) * GA_LDR + ((idx & 7) << 2)) = m;
                if (REAL) *reinterpret_cast<half4*>(mk + (NB * 32 + (idx >> 3)) * GA_LDR + ((idx & 7) << 2)) = ml;
            }
        }
    };

    f32x16 acc[NB];
    f32x16 accc;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accc[r] = 0.f;
    half8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.f;

    const bool has_cb = (wave * 32 < C);  // this wave owns channel block `wave` (uniform)
    const bool has_cnt = (wave < NB);     // this wave counts pixels of n-block `wave`

    // `odd32`: the 32-px tile is the second half of its 64-px tile (BITS: selects bits 16.. of the even / odd words)
    auto compute = [&](int buf, int odd32) {
        const _Float16* xh = lds + (size_t)buf * rows_buf * GA_LDR;
        const _Float16* xl = xh + C * GA_LDR;
        const _Float16* mk = xl + C * GA_LDR;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = (ks << 4) + (g << 3);
            const int bsh = 16 * odd32 + 8 * ks + 4 * g;  // pixels 16 ks + 8 g + e  <->  bit bsh + e / 2 of the even (e even) / odd word
            if (has_cb) {
                const half8 bh = *reinterpret_cast<const half8*>(xh + (wave * 32 + li) * GA_LDR + off);
                const half8 bl = *reinterpret_cast<const half8*>(xl + (wave * 32 + li) * GA_LDR + off);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    half8 a;
                    if (BITS == 1) {
                        const unsigned we = reinterpret_cast<const unsigned*>(mk)[nb * 32 + li];
                        const unsigned wo = reinterpret_cast<const unsigned*>(mk)[NB * 32 + nb * 32 + li];
                        a = lut[((we >> bsh) & 0xFu) | (((wo >> bsh) & 0xFu) << 4)];
                    } else {
                        a = *reinterpret_cast<const half8*>(mk + (nb * 32 + li) * GA_LDR + off);
                    }
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc[nb], 0, 0, 0);
                    if (!XH) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[nb], 0, 0, 0);
                    if (REAL) {  // a = a_hi here: + a_lo * x_hi (a_lo * x_lo is below fp32 resolution)
                        const half8 al = *reinterpret_cast<const half8*>(mk + (NB * 32 + nb * 32 + li) * GA_LDR + off);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nb], 0, 0, 0);
                    }
                }
            }
            if (has_cnt) {
                half8 a;
                if (BITS == 1) {
                    const unsigned we = reinterpret_cast<const unsigned*>(mk)[wave * 32 + li];
                    const unsigned wo = reinterpret_cast<const unsigned*>(mk)[NB * 32 + wave * 32 + li];
                    a = lut[((we >> bsh) & 0xFu) | (((wo >> bsh) & 0xFu) << 4)];
                } else {
                    a = *reinterpret_cast<const half8*>(mk + (wave * 32 + li) * GA_LDR + off);
                }
                accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, accc, 0, 0, 0);
                if (REAL) {  // `cnt` = sum_p a (hi + lo)
                    const half8 al = *reinterpret_cast<const half8*>(mk + (NB * 32 + wave * 32 + li) * GA_LDR + off);
                    accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ones, accc, 0, 0, 0);
                }
            }
        }
    };

    // pipeline over the FULL tiles: tile t computes from LDS[t&1]; tile t+1 sits in one register set and is committed to
    // LDS[(t+1)&1] after the MFMAs; tile t+2 is issued into the other register set before them (two tiles = 96 KB per CU in
    // flight).  Issues past the end re-read the last full tile (clamped) so every iteration has the same 6 loads.
    const int nfull = ileave ? ntiles : (vec_ok ? (p_end - p_begin) / GA_PT : 0);
    if (nfull > 0) {
        issue(0, xrA, mrA);
        issue(min(1, nfull - 1), xrB, mrB);
        commit(0, xrA, mrA);
        __syncthreads();
        for (int t = 0; t < nfull; t += 2) {
            issue(min(t + 2, nfull - 1), xrA, mrA);
            compute(0, (tile_p0(t) >> 5) & 1);
            if (t + 1 < nfull) commit(1, xrB, mrB);
            __syncthreads();
            if (t + 1 >= nfull) break;
            issue(min(t + 3, nfull - 1), xrB, mrB);
            compute(1, (tile_p0(t + 1) >> 5) & 1);
            if (t + 2 < nfull) commit(0, xrA, mrA);
            __syncthreads();
        }
    }
    for (int t = nfull; t < ntiles; ++t) {  // ragged / unaligned tiles
        issue_slow(t, xrA, mrA);
        commit(0, xrA, mrA);
        __syncthreads();
        compute(0, 0);
        __syncthreads();
    }

    // ---- write this workgroup's partial (rows of the n-chunk, zero when the range was empty)
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;
    if (has_cb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                pp[(size_t)n * C + wave * 32 + li] = acc[nb][r];
            }
    }
    if (has_cnt && li == 0) {
        float* cp = cntp + ((size_t)b * G + gidx) * NPT;
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[n0 + wave * 32 + vkn_cd_row(r, lane)] = accc[r];
    }
}

// ---- wave-independent variant for the bit-word operand (stages > 0 of the fused head).
// k_gather_mfma synchronises its 8 waves once per 32-px tile, so all of them split/write LDS at the same time and then all run
// MFMAs at the same time (PMC: waves wait 54 % of their cycles, MFMA pipe 5 % busy).  With the binary operand available as words
// nothing has to be shared between waves: wave w streams ONLY its 32 channels of x (4 KB per 32-px tile), transposes them through
// a PRIVATE LDS tile and takes the mask fragments from the (even nibble, odd nibble) table.  No workgroup barrier in the loop
// (LDS operations of one wave execute in order), waves drift apart and their load / VALU / MFMA phases overlap.
// Work unit = one 64-px super-tile (both 32-px halves share the even / odd words).  Same partial layout as k_gather_mfma.
template <int NB>
__global__ __launch_bounds__(GA_THREADS, 2) void k_gather_bits_w(const float* __restrict__ x, const unsigned* __restrict__ bits,
                                                                  float* __restrict__ part, float* __restrict__ cntp, int N,
                                                                  int NPT, int n0, int C, int P, int px_per_wg, int ileave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WTILE = 2 * 32 * GA_LDR;  // halfs of one private tile: hi [32][40] + lo [32][40]
    const int b = blockIdx.y, gidx = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    _Float16* wl = reinterpret_cast<_Float16*>(smem) + (size_t)wave * 2 * WTILE;  // two tiles (the halves of a super-tile)
    half8* lut = reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(smem) + (size_t)GA_WAVES * 2 * WTILE);
    for (int v = tid; v < 256; v += GA_THREADS) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = ((v >> ((e >> 1) + 4 * (e & 1))) & 1) ? (_Float16)1.f : (_Float16)0.f;
        lut[v] = h;
    }
    __syncthreads();  // the only workgroup barrier

    // super-tile s of this workgroup starts at pixel sup_p0(s) (same two mappings as k_gather_mfma)
    const int p_begin = gidx * px_per_wg;
    const int p_end = min(P, p_begin + px_per_wg);
    const int nsup = ileave ? (((P >> 6) - gidx + G - 1) / G)
                            : ((p_end > p_begin) ? (p_end - p_begin) >> 6 : 0);  // launcher: P % 64 == 0, px_per_wg % 64 == 0
    auto sup_p0 = [&](int s) { return ileave ? ((s * G + gidx) << 6) : p_begin + (s << 6); };
    const bool has_cb = (wave * 32 < C);
    const bool has_cnt = (wave < NB);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    unsigned cnt_i = 0;  // ON pixels of row (n-block `wave`, row li): popcount of its words (exact; no ones-MFMA needed here)

    if (has_cb && nsup > 0) {
        const float* xw = x + ((size_t)b * C + wave * 32) * P;
        const unsigned* wb = bits + (size_t)b * (P >> 5) * NPT + n0 + li;
        // lane's slots in a 32-channel x 32-px tile: rows (lane >> 3) + 8 i, 16-byte chunk lane & 7
        const int lrow = lane >> 3, lchk = lane & 7;
        f32x4 xa[8], xb[8];        // a super-tile = 2 tiles x 4 dwordx4 per lane
        unsigned wa[2 * NB], wv[2 * NB];

        auto issue = [&](int s, f32x4 (&xr)[8], unsigned (&wr)[2 * NB]) {
            const int p0 = sup_p0(s);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xr[h * 4 + i] = __builtin_nontemporal_load(
                        reinterpret_cast<const f32x4*>(xw + (size_t)(lrow + 8 * i) * P + p0 + 32 * h + (lchk << 2)));
            const unsigned* wp = wb + (size_t)((p0 >> 6) << 1) * NPT;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                wr[nb] = wp[nb * 32];
                wr[NB + nb] = wp[NPT + nb * 32];
            }
        };
        auto commit = [&](const f32x4 (&xr)[8]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                _Float16* xh = wl + h * WTILE;
                _Float16* xl = xh + 32 * GA_LDR;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    half4 hh, ll;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        _Float16 a, c2;
                        vkn_split_f16(xr[h * 4 + i][k], a, c2);
                        hh[k] = a;
                        ll[k] = c2;
                    }
                    const int off = (lrow + 8 * i) * GA_LDR + (lchk << 2);
                    *reinterpret_cast<half4*>(xh + off) = hh;
                    *reinterpret_cast<half4*>(xl + off) = ll;
                }
            }
        };
        auto compute = [&](const unsigned (&wr)[2 * NB]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const _Float16* xh = wl + h * WTILE;
                const _Float16* xl = xh + 32 * GA_LDR;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = (ks << 4) + (g << 3);
                    const int bsh = 16 * h + 8 * ks + 4 * g;
                    const half8 bh = *reinterpret_cast<const half8*>(xh + li * GA_LDR + off);
                    const half8 bl = *reinterpret_cast<const half8*>(xl + li * GA_LDR + off);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const half8 a = lut[((wr[nb] >> bsh) & 0xFu) | (((wr[NB + nb] >> bsh) & 0xFu) << 4)];
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[nb], 0, 0, 0);
                    }
                }
            }
        };
        auto lds_fence = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes / reads have completed
            __builtin_amdgcn_wave_barrier();
        };

        const int last = nsup - 1;  // two register sets (a third one spills: 141 us instead of 71)
        issue(0, xa, wa);
        for (int s = 0; s < nsup; s += 2) {
            issue(min(s + 1, last), xb, wv);
            commit(xa);
            lds_fence();
            compute(wa);
            lds_fence();
            if (s + 1 >= nsup) break;
            issue(min(s + 2, last), xa, wa);
            commit(xb);
            lds_fence();
            compute(wv);
            lds_fence();
        }
    }

    if (has_cnt) {  // wave w < NB: ON-pixel count of the rows of n-block w over this workgroup's range
        const unsigned* wc = bits + (size_t)b * (P >> 5) * NPT + n0 + wave * 32 + li;
        for (int s2 = 0; s2 < nsup; ++s2) {
            const unsigned* wp = wc + (size_t)((sup_p0(s2) >> 6) << 1) * NPT;
            cnt_i += __popc(wp[0]) + __popc(wp[NPT]);
        }
    }

    // ---- write this workgroup's partial (rows of the n-chunk, zero when the range was empty)
    float* pp = part + ((size_t)b * G + gidx) * NPT * C;
    if (has_cb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + nb * 32 + vkn_cd_row(r, lane);
                pp[(size_t)n * C + wave * 32 + li] = acc[nb][r];
            }
    }
    if (has_cnt && g == 0) cntp[((size_t)b * G + gidx) * NPT + n0 + wave * 32 + li] = (float)cnt_i;
}

// xraw[b][n][c] = sum_g part[b][g][n][c], cnt[b][n] likewise.  Fixed summation tree: four interleaved running sums over g
// (chain k = groups k, k+4, ...; a remainder of G % 4 groups continues chain 0), combined as (s0 + s1) + (s2 + s3) -> deterministic.
// One wave per chain, 8 independent 16-byte loads in flight per lane: with few frames G is large (256 partials per frame at B = 1)
// and a serial walk over g is pure load latency (73 us per call at B = 1 before this layout; the sums are bit-identical to it).
// `status` (or NULL): VKN_STATUS_RANGE is OR-ed in when a gathered sum is not finite — what |x| >= 65504 or a non-finite x turns
// into once it has been through the f16 split (inf x 0 = NaN, inf x 1 = inf: EVERY kernel row of the frame sees the pixel), so one
// compare per OUTPUT value of this small kernel watches the whole feature map for free (include/vkn.h: VKN_E_RANGE).
__global__ __launch_bounds__(256) void k_gather_reduce(const float* __restrict__ part, const float* __restrict__ cntp,
                                                        float* __restrict__ xraw, float* __restrict__ cnt, int N, int NPT,
                                                        int C, int G, int* __restrict__ status, const char* __restrict__ touch,
                                                        unsigned touch_bytes) {
    // `touch` (or NULL): the pre-split weights of the [N x C] chain that runs right behind this kernel.  Between two stages the
    // x-streaming passes push > 1 GB through the memory-side cache, so the chain's 117 workgroups — all streaming the same 12 MB in
    // lock-step — would take every tile's first touch from HBM; this latency-bound kernel has the bandwidth to spare and pulls them
    // into the memory-side cache on the way (one 16-byte load per thread and 64 KB step; the values are discarded).
    if (touch) {
        const unsigned per = (touch_bytes / gridDim.x + 15u) & ~15u;
        const unsigned t0 = blockIdx.x * per, t1 = min(t0 + per, touch_bytes & ~15u);
        for (unsigned o = t0 + threadIdx.x * 16u; o < t1; o += 256u * 16u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(touch + o);
            asm volatile("" ::"v"(v));
        }
    }
    __shared__ f32x4 comb[3][64];
    const int row = blockIdx.x;  // b*N + n
    const int b = row / N, n = row - b * N;
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // chain
    const size_t gstride = (size_t)NPT * C;
    const float* pp = part + ((size_t)b * G * NPT + n) * C;
    const int G4 = G & ~3;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c4 = c0 + lane * 4;
        const bool ok = c4 < C;
        const float* pc = pp + (ok ? c4 : 0);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int gi = k;
        for (; gi + 28 < G4; gi += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(pc + (size_t)(gi + 4 * u) * gstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; gi < G4; gi += 4) s += *reinterpret_cast<const f32x4*>(pc + (size_t)gi * gstride);
        if (k == 0)
            for (gi = G4; gi < G; ++gi) s += *reinterpret_cast<const f32x4*>(pc + (size_t)gi * gstride);
        if (c0) __syncthreads();
        if (k) comb[k - 1][lane] = s;
        __syncthreads();
        if (k == 0 && ok) {
            const f32x4 r = (s + comb[0][lane]) + (comb[1][lane] + comb[2][lane]);
            *reinterpret_cast<f32x4*>(xraw + (size_t)row * C + c4) = r;
            if (status && !(fabsf(r[0]) <= 3.0e38f && fabsf(r[1]) <= 3.0e38f && fabsf(r[2]) <= 3.0e38f && fabsf(r[3]) <= 3.0e38f))
                atomicOr(status, 1);   // VKN_STATUS_RANGE (rare: no contention)
        }
    }
    if (k == 1) {  // lane l sums groups l, l + 64, ... in order, then a fixed butterfly (binary counts: exact in any order)
        const float* cp = cntp + (size_t)b * G * NPT + n;
        float s = 0.f;
        for (int gi = lane; gi < G; gi += 64) s += cp[(size_t)gi * NPT];
        s = vkn_wave_sum(s);
        if (lane == 0) cnt[row] = s;
    }
}

// Exact-fp32 debug / fallback: one workgroup per (b, n), threads over channels, p-ordered accumulation.
__global__ __launch_bounds__(256) void k_gather_ref(const float* __restrict__ x, const float* __restrict__ masks,
                                                    float thr, float* __restrict__ xraw, float* __restrict__ cnt, int N,
                                                    int C, int P, long long mask_fs) {
    const int n = blockIdx.x, b = blockIdx.y;
    const float* mp = masks + (size_t)b * mask_fs + (size_t)n * P;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float* xp = x + ((size_t)b * C + c) * P;
        float s = 0.f;
        for (int p = 0; p < P; ++p)
            if (mp[p] >= thr) s += xp[p];
        xraw[((size_t)b * N + n) * C + c] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int p = 0; p < P; ++p)
            if (mp[p] >= thr) s += 1.f;
        cnt[(size_t)b * N + n] = s;
    }
}

int vkn_launch_gather_reduce(const float* part, const float* cntp, float* xraw, float* cnt, int B, int N, int C, int G,
                             hipStream_t stream, int* status, const void* touch, size_t touch_bytes) {
    if (touch_bytes >= (1ull << 31)) touch = nullptr;
    hipLaunchKernelGGL(k_gather_reduce, dim3(B * N), dim3(256), 0, stream, part, cntp, xraw, cnt, N, (N + 31) / 32 * 32, C, G, status,
                       static_cast<const char*>(touch), (unsigned)touch_bytes);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// number of per-frame partials the launcher will produce for (B, P) — workspace sizing
int vkn_gather_groups(int B, int P) {
    int wg_per_frame = 256 / (B > 0 ? B : 1);
    if (wg_per_frame < 1) wg_per_frame = 1;
    int px_per_wg = (P + wg_per_frame - 1) / wg_per_frame;
    px_per_wg = (px_per_wg + GA_PT - 1) / GA_PT * GA_PT;
    return (P + px_per_wg - 1) / px_per_wg;
}

// part: [B][G][NPT][C] f32, cntp: [B][G][NPT] f32 (workspace); xraw [B][N][C], cnt [B][N] outputs.
int vkn_launch_gather(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp,
                      int B, int N, int C, int P, hipStream_t stream, int xdt, int* status, const void* touch, size_t touch_bytes) {
    return vkn_launch_gather_ex(x, masks, thr, xraw, cnt, part, cntp, B, N, C, P, N, stream, xdt, status, touch, touch_bytes);
}

// mask_rows: rows per frame of the logits tensor the N gathered rows live in (>= N; `masks` points at the first of them)
static int gather_launch(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp, int B,
                         int N, int C, int P, int mask_rows, int bits, hipStream_t stream, int xdt = 0, int* status = nullptr,
                         const void* touch = nullptr, size_t touch_bytes = 0, int x_alias = -1);

int vkn_launch_gather_ex(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp,
                         int B, int N, int C, int P, int mask_rows, hipStream_t stream, int xdt, int* status, const void* touch,
                         size_t touch_bytes) {
    return gather_launch(x, masks, thr, xraw, cnt, part, cntp, B, N, C, P, mask_rows, 0, stream, xdt, status, touch, touch_bytes);
}

// REAL-valued left operand a [B][mask_rows][P] (first N rows used): xraw = sum_p a x, cnt = sum_p a
// x_alias >= 0: channels c >= x_alias of x are max(x[c - x_alias], x_floor) and are not stored (x holds x_alias rows — or C when x_alias == 0:
// every channel floored); -1: off
int vkn_launch_gather_real(const float* x, const float* a, float* xraw, float* cnt, float* part, float* cntp, int B, int N, int C,
                           int P, int mask_rows, hipStream_t stream, int x_alias, float x_floor) {
    if (x_alias > C || (x_alias > 0 && (C - x_alias > x_alias || B != 1))) return VKN_E_ARG;   // (aliased rows: one frame — the frame stride is C rows)
    return gather_launch(x, a, x_alias >= 0 ? x_floor : 0.f, xraw, cnt, part, cntp, B, N, C, P, mask_rows, 2, stream, 0, nullptr, nullptr, 0,
                         x_alias);
}

// soft gather weights: xraw = sum_p [z >= thr] sigmoid(z) x, cnt = the sum of the weights (use_binary=False, knet/det/kernel_head.py:243-249)
int vkn_launch_gather_soft(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp, int B,
                           int N, int C, int P, int mask_rows, hipStream_t stream) {
    return gather_launch(x, masks, thr, xraw, cnt, part, cntp, B, N, C, P, mask_rows, 3, stream);
}

// binary operand given as bit words [B][P/64][2][roundup(N,32)] (vkn_launch_decode_bits); P % 64 == 0
int vkn_launch_gather_bits(const float* x, const unsigned* bits, float* xraw, float* cnt, float* part, float* cntp, int B, int N,
                           int C, int P, hipStream_t stream, int xdt, int* status, const void* touch, size_t touch_bytes) {
    if ((P % 64) != 0) return VKN_E_SHAPE;
    return gather_launch(x, reinterpret_cast<const float*>(bits), 0.f, xraw, cnt, part, cntp, B, N, C, P, N, 1, stream, xdt, status, touch,
                         touch_bytes);
}

static int gather_launch(const float* x, const float* masks, float thr, float* xraw, float* cnt, float* part, float* cntp, int B,
                         int N, int C, int P, int mask_rows, int bits, hipStream_t stream, int xdt, int* status, const void* touch,
                         size_t touch_bytes, int x_alias) {
    if (x_alias >= 0 && bits != 2) return VKN_E_ARG;
    if (B <= 0 || N <= 0 || P <= 0 || mask_rows < N) return VKN_E_ARG;
    if (xdt < 0 || xdt > 2) return VKN_E_ARG;
    if (xdt && (bits >= 2 || (P % 64) != 0)) return VKN_E_SHAPE;  // half-storage x: binary operands, whole 16-byte aligned tiles
    const long long mask_fs = (long long)mask_rows * P;
    if (C % 32 != 0 || C > 256) return VKN_E_SHAPE;
    const int NPT = (N + 31) / 32 * 32;
    int wg_per_frame = 256 / B;
    if (wg_per_frame < 1) wg_per_frame = 1;
    int px_per_wg = (P + wg_per_frame - 1) / wg_per_frame;
    px_per_wg = (px_per_wg + GA_PT - 1) / GA_PT * GA_PT;
    const int G = (P + px_per_wg - 1) / px_per_wg;
    const int ileave = ((P % 64) == 0 && vkn_dbg_env("VKN_GATHER_ILEAVE", 1) != 0) ? 1 : 0;
    const bool wave_indep = bits == 1 && xdt == 0 && vkn_dbg_env("VKN_GATHER_BITS_W", 1) != 0 && (C % 32) == 0 &&
                            (px_per_wg % 64) == 0;
    for (int n0 = 0; n0 < NPT; n0 += 128) {
        const int nb = (NPT - n0 >= 128) ? 4 : (NPT - n0) / 32;
        dim3 grid(G, B, 1), block(GA_THREADS);
        if (wave_indep) {
            const size_t ldsw = (size_t)GA_WAVES * 2 * 2 * 32 * GA_LDR * sizeof(_Float16) + 4096;
            const unsigned* bw = reinterpret_cast<const unsigned*>(masks);
#define GA_WLAUNCH(NBV)                                                                                              \
    case NBV:                                                                                                        \
        VKN_ALLOW_FULL_LDS(k_gather_bits_w<NBV>);                                                                    \
        hipLaunchKernelGGL(k_gather_bits_w<NBV>, grid, block, ldsw, stream, x, bw, part, cntp, N, NPT, n0, C, P, px_per_wg, ileave); \
        break;
            switch (nb) {
                GA_WLAUNCH(1)
                GA_WLAUNCH(2)
                GA_WLAUNCH(3)
                GA_WLAUNCH(4)
                default:
                    return VKN_E_SHAPE;
            }
#undef GA_WLAUNCH
            VKN_CHECK_LAUNCH();
            continue;
        }
        const size_t lds = (size_t)2 * (2 * C + (bits >= 2 ? 2 : 1) * nb * 32) * GA_LDR * sizeof(_Float16) + (bits == 1 ? 4096 : 0);
#define GA_LAUNCH(NBV, BV, XHV)                                                                                  \
    do {                                                                                                         \
        VKN_ALLOW_FULL_LDS((k_gather_mfma<NBV, BV, XHV>));                                                       \
        hipLaunchKernelGGL((k_gather_mfma<NBV, BV, XHV>), grid, block, lds, stream, x, masks, thr, part, cntp, N, NPT, n0, C, P, \
                           px_per_wg, mask_fs, ileave, x_alias);                                                 \
    } while (0)
#define GA_CASE(NBV)                                 \
    case NBV:                                        \
        if (bits == 3) GA_LAUNCH(NBV, 3, 0);         \
        else if (bits == 2) GA_LAUNCH(NBV, 2, 0);    \
        else if (bits && xdt == 1) GA_LAUNCH(NBV, 1, 1); \
        else if (bits && xdt == 2) GA_LAUNCH(NBV, 1, 2); \
        else if (bits) GA_LAUNCH(NBV, 1, 0);         \
        else if (xdt == 1) GA_LAUNCH(NBV, 0, 1);     \
        else if (xdt == 2) GA_LAUNCH(NBV, 0, 2);     \
        else GA_LAUNCH(NBV, 0, 0);                   \
        break;
        switch (nb) {
            GA_CASE(1)
            GA_CASE(2)
            GA_CASE(3)
            GA_CASE(4)
            default:
                return VKN_E_SHAPE;
        }
#undef GA_CASE
#undef GA_LAUNCH
        VKN_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_gather_reduce, dim3(B * N), dim3(256), 0, stream, part, cntp, xraw, cnt, N, NPT, C, G, status,
                       static_cast<const char*>(touch), (unsigned)(touch_bytes < (1ull << 31) ? touch_bytes : 0));
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_gather_ref(const float* x, const float* masks, float thr, float* xraw, float* cnt, int B, int N, int C,
                          int P, hipStream_t stream) {
    return vkn_launch_gather_ref_ex(x, masks, thr, xraw, cnt, B, N, C, P, N, stream);
}

int vkn_launch_gather_ref_ex(const float* x, const float* masks, float thr, float* xraw, float* cnt, int B, int N, int C,
                             int P, int mask_rows, hipStream_t stream) {
    if (mask_rows < N) return VKN_E_ARG;
    hipLaunchKernelGGL(k_gather_ref, dim3(N, B), dim3(256), 0, stream, x, masks, thr, xraw, cnt, N, C, P,
                       (long long)mask_rows * P);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
