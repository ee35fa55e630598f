// vkn_chain.hip — the [B*N, C] "kernel update + interaction" chain of one stage as TWO persistent row-owner kernels around the
// attention (C == 256, the shipped shape).  Replaces the ten k_gemm_s3 / k_ffn_fused / k_rowepi launches per stage of vkn_update.hip
// on the fast path; that file keeps the generic path (any C, exact-fp32 GEMMs, split-K) and the link blocks.
//
// Covers (reference file:line):
//   KernelUpdator.forward                      knet/kernel_updator.py:56-93                 -> k_chain_a (+ in_proj of the attention)
//   attention_norm(attention(obj))             knet/det/kernel_update_head.py:204-208       in_proj: k_chain_a; softmax(QK^T)V: k_attn_mfma;
//                                                                                           out_proj + residual + LayerNorm: k_chain_c
//   ffn_norm(ffn(obj))                         knet/det/kernel_update_head.py:214-215       -> k_chain_c
//   cls_fcs / fc_cls / mask_fcs / fc_mask      knet/det/kernel_update_head.py:217-227       -> k_chain_c (fc_mask folded with feat_transform)
//
// Design (DESIGN.md §5 "chain"):
//   * A workgroup (8 waves) OWNS 32 rows (kernels) for the whole launch.  Activations never leave the CU between GEMMs: they live in
//     LDS as bf16x3 "images" [plane 3][row 32][k 256] (16-byte chunk j of row r stored at chunk j ^ r: conflict-free ds_read_b128) and,
//     where a later epilogue needs them elementwise, in registers.
//   * Weights are the pre-split bf16x3 tile images of vkn_prepare_stage_f32 ([plane][q][row 256][8] per 256-col x 32-k tile = already
//     the MFMA fragment layout).  Wave w is the only consumer of column block w, so every wave streams ITS six fragments per K-tile
//     straight from global memory (L2 / MALL resident: 12 MB per stage, read by all workgroups in near lock-step) into a 4-deep
//     register ring with buffer loads — no LDS traffic for weights, no barrier inside a GEMM, and the stream never stops: the ring is
//     refilled with the NEXT GEMM's first tiles while the current epilogue runs (the address sequence is static).
//   * MFMA operand roles are swapped against k_gemm_s3: D^T = W . A^T, i.e. the weight fragment is the first operand and the
//     activation fragment the second.  A lane then owns ONE activation row (lane & 31) and sixteen output columns in four groups of four
//     consecutive columns: LayerNorm needs a 16-value in-lane sum + one exchange of 16 partials per row through 2 KB of LDS (no
//     output tile round trip, no per-row wave reductions), every elementwise combination of the updator (gates, mix) is a register
//     operation, image writes are 8-byte packed stores and global stores are float4.
//   * Six significant cross products of the bf16x3 split per operand pair, fp32 accumulation, smallest terms first — the arithmetic
//     of k_gemm_s3 (2^-24 relative); LayerNorm two-pass in fp32.  Summation order differs from the launch-per-GEMM path (the
//     LayerNorm partials, the FFN's hidden chunks in sequence instead of four split sums): results agree to fp32 rounding, not bit
//     for bit; everything is deterministic.
#include "vkn_common.h"
#include "vkn_launch.h"

// This file compiles TWICE: as itself (the three-term bf16 split: 6 bytes and 6 MFMAs per operand pair, 2^-24) and, with CH_H2 defined, from
// vkn_chain_h2.hip (the TWO-term fp16 split hi + lo of the gather / decode kernels: 4 bytes, 3 MFMAs: what the persistent form runs unless VKN_FLAG_CHAIN_BF16X3).  Everything that
// depends on the split — element type, planes per image / tile, the products — hangs on the macros below; see vkn_chain_h2.hip for the
// range management of the fp16 form (pre-scaled weight images, per-row scaled activation images of unbounded rows).
#ifdef CH_H2
typedef _Float16 ch_e;
#define CH_NPL 2                        // planes per image / weight tile
#define CH_WTILE 32768u                 // bytes of one weight tile image (256 cols x 32 k x 2 planes fp16)
#define k_chain_a k_chain_a_h2
#define k_chain_c k_chain_c_h2
#define ChainAArgs ChainAArgsH2
#define ChainCArgs ChainCArgsH2
#define vkn_launch_chain_a vkn_launch_chain_a_h2
#define vkn_launch_chain_c vkn_launch_chain_c_h2
#else
typedef __bf16 ch_e;
#define CH_NPL 3
#define CH_WTILE 49152u                 // bytes of one weight tile image (256 cols x 32 k x 3 planes bf16)
#endif
typedef ch_e ch_ex8 __attribute__((ext_vector_type(8)));
typedef ch_e ch_ex4 __attribute__((ext_vector_type(4)));
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));

#define CH_THREADS 512
#define CH_ROWS 32
#define CH_C 256
#define CH_IMG (CH_NPL * CH_ROWS * CH_C)   // elements of one activation image (48 KB; fp16 form 32 KB)
#define CH_PLANE (CH_ROWS * CH_C)       // elements of one plane of an image
#define CH_SBUF (2 * 16 * 32)           // floats of one row-statistics exchange buffer: [value 2][partial 16][row 32]

// this wave's LDS operations are done + workgroup barrier, as ONE asm statement: behind a __syncthreads() hipcc strengthens the wait
// to vmcnt(0) (workgroup release fence), which would drain the weight ring at every epilogue
#define CH_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {

// v = p[0] + p[1] (+ p[2]): each term the rounding of what the earlier ones left
__device__ __forceinline__ void ch_split(float v, ch_e (&p)[CH_NPL]) {
    float r = v;
#pragma unroll
    for (int i = 0; i < CH_NPL; ++i) {
        p[i] = (ch_e)r;
        if (i + 1 < CH_NPL) r -= (float)p[i];
    }
}
// four values -> CH_NPL packed 8-byte plane chunks
__device__ __forceinline__ void ch_split4(const float (&v)[4], ch_ex4 (&o)[CH_NPL]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ch_e t[CH_NPL];
        ch_split(v[e], t);
#pragma unroll
        for (int i = 0; i < CH_NPL; ++i) o[i][e] = t[i];
    }
}
// the power of two that puts a row maximum m at [2^9, 2^10) — the fp16 form's per-row scale of activation rows whose magnitude the
// chain does not bound (gather sums, incoming kernels, the updator's gate product, attention output); s = 1 for an all-zero row
__device__ __forceinline__ float ch_row_pow2(float m, float& inv) {
    int e = 0;
    if (m > 0.f) (void)frexpf(m, &e); else e = 10;
    const int sh = min(max(10 - e, -60), 60);
    inv = ldexpf(1.0f, -sh);
    return ldexpf(1.0f, sh);
}

// lane coordinates of a workgroup thread
struct ChLane {
    int wave, g, li;
    unsigned woff;   // byte offset of this lane's fragment inside a weight tile image: plane 0, ks 0
    unsigned abase;  // byte offset of this lane's activation fragment inside an image plane for chunk pair 0: row li, 16-byte chunk
                     // (g ^ li); the fragment of (kt, ks) is at abase ^ ((4 kt + 2 ks) << 4) — the swizzle is an XOR of chunk bits
    int* status;     // two-term fp16 form: where ch_img_write reports an activation outside the fp16 envelope (or NULL)
};
__device__ __forceinline__ ChLane ch_lane(int tid) {
    ChLane L;
    const int lane = tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.g = lane >> 5;
    L.li = lane & 31;
    L.woff = (unsigned)(L.g * 4096 + (L.wave * 32 + L.li) * 16);
    L.abase = (unsigned)(L.li * (CH_C * 2) + (((L.g ^ L.li) & 31) << 4));
    L.status = nullptr;
    return L;
}

// ring depth and cache policy: 4 units, default policy; the debug build's stream probes (VKN_CHAIN_ABL 5 / 6 / 7, all without MFMAs)
// change them to tell a latency-bound weight stream from a bandwidth-bound one
#ifdef VKN_DEBUG
#define CH_RING_OF(ABL) ((ABL) == 5 ? 2 : (ABL) == 6 ? 8 : 4)
#define CH_AUX_OF(ABL) ((ABL) == 7 ? 2 : (ABL) == 8 ? 16 : (ABL) == 9 ? 17 : (ABL) == 10 ? 1 : 0)   /* 8 / 9 / 10: the FULL kernel with sc1 / sc0 sc1 / sc0 loads */
#define CH_NOMFMA(ABL) ((ABL) == 1 || (ABL) == 4 || (ABL) == 5 || (ABL) == 6 || (ABL) == 7)
#define CH_NOLOAD(ABL) ((ABL) == 2 || (ABL) == 4)
#ifdef CH_H2
#define CH_NP_OF(ABL) 2
#else
#define CH_NP_OF(ABL) ((ABL) == 11 ? 2 : 3)   /* 11: the cost of a two-term split (4 B / weight, 3 products): the prize of VERDICT r04 item 4 (built since: CH_H2) */
#endif
#else
#define CH_NP_OF(ABL) CH_NPL
#define CH_RING_OF(ABL) 4
#define CH_AUX_OF(ABL) 0
#define CH_NOMFMA(ABL) false
#define CH_NOLOAD(ABL) false
#endif
template <int RING>
struct ChRing {
    cu32x4 r[RING][2 * CH_NPL];  // [slot][ks * CH_NPL + plane]
};

// the six fragments (2 k-steps x 3 planes) of this wave's column block of the tile at byte offset `toff` of the weight buffer
// NP (debug build, VKN_CHAIN_ABL 11): planes loaded per fragment — 2 = the traffic of a two-term (4 bytes / weight) split
template <int RING, int AUX = 0, int NP = CH_NPL>
__device__ __forceinline__ void ch_wload(ChRing<RING>& R, const int slot, const __amdgpu_buffer_rsrc_t wrs, const ChLane& L, unsigned toff) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < NP; ++p)
            R.r[slot][ks * CH_NPL + p] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)L.woff, (int)(toff + (unsigned)((p * 4 + 2 * ks) * 4096)), AUX);
}

// activation fragments (CH_NPL planes) of K-tile kt, k-step ks from an image
__device__ __forceinline__ void ch_afrag(const ch_e* img, const ChLane& L, int kt, int ks, ch_ex8 (&a)[CH_NPL]) {
    const char* pb = reinterpret_cast<const char*>(img) + (L.abase ^ (unsigned)(((kt << 2) + (ks << 1)) << 4));
    const ch_e* p = reinterpret_cast<const ch_e*>(pb);
#pragma unroll
    for (int i = 0; i < CH_NPL; ++i) a[i] = *reinterpret_cast<const ch_ex8*>(p + i * CH_PLANE);
}

// acc (transposed tile: lane = activation row li, register r = output column 8 (r >> 2) + 4 g + (r & 3) of the wave's block) +=
// W-fragments (slot) x activation fragments: the significant cross products, smallest first
template <int RING>
__device__ __forceinline__ void ch_mfma(f32x16& acc, const ChRing<RING>& R, const int slot, const int ks, const ch_ex8 (&a)[CH_NPL]) {
#ifdef CH_H2
    const ch_ex8 wh = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 2 + 0]);
    const ch_ex8 wl = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 2 + 1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, a[0], acc, 0, 0, 0);   // lo x lo is below fp32 resolution
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, a[0], acc, 0, 0, 0);
#else
    const ch_ex8 wh = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 3 + 0]);
    const ch_ex8 wm = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 3 + 1]);
    const ch_ex8 wl = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 3 + 2]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, a[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, a[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a[0], acc, 0, 0, 0);
#endif
}
#if defined(VKN_DEBUG) && !defined(CH_H2)
// (debug build, VKN_CHAIN_ABL 11: WRONG results by construction — bf16 x 2 precision) the three products of a two-term split
template <int RING>
__device__ __forceinline__ void ch_mfma3(f32x16& acc, const ChRing<RING>& R, const int slot, const int ks, const ch_ex8 (&a)[CH_NPL]) {
    const ch_ex8 wh = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 3 + 0]);
    const ch_ex8 wm = __builtin_bit_cast(ch_ex8, R.r[slot][ks * 3 + 1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, a[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a[0], acc, 0, 0, 0);
}
#endif

// what the ring is refilled with once the current GEMM has no tiles left to request: the first three units of the next GEMM
struct ChNext {
    unsigned base0, base1;  // tile-image byte offsets of the next GEMM's accumulator 0 / 1 (column tiles or weights)
    int nacc;               // 1 or 2; 0 = nothing follows
    __device__ __forceinline__ unsigned off(int j) const {  // unit j (0 .. RING-2) of the next GEMM
        if (nacc == 2) return ((j & 1) ? base1 : base0) + (unsigned)(j >> 1) * CH_WTILE;
        return base0 + (unsigned)j * CH_WTILE;
    }
};

// One GEMM of the chain over K = 256 (8 K-tiles): NACC accumulators (column tiles or independent weights) fed from the unit stream
//   unit u = kt * NACC + a  ->  tile image at base[a] + kt * CH_WTILE, accumulator a, activation image img[a].
// The ring slot of unit u is u % RING (every GEMM's unit count is a multiple of RING); on entry units 0 .. RING-2 are in flight
// (requested by the previous GEMM's tail or by the kernel prologue), on exit the next GEMM's first RING-1 units are.
// ABL (debug build only, VKN_CHAIN_ABL; WRONG results by construction, time attribution): 1 = no MFMAs, 2 = no weight loads,
// 3 = no activation-fragment reads, 4 = neither MFMAs nor loads (epilogues alone), 5 / 6 / 7 = no MFMAs with a ring of 2 / 8 units /
// non-temporal loads
template <int NACC, bool SAMEA, int ABL>
__device__ __forceinline__ void ch_gemm(f32x16 (&acc)[NACC], const ch_e* img0, const ch_e* img1, unsigned base0, unsigned base1,
                                        const ChNext nx, ChRing<CH_RING_OF(ABL)>& R, const __amdgpu_buffer_rsrc_t wrs, const ChLane& L) {
    constexpr int NU = 8 * NACC, RING = CH_RING_OF(ABL), AUX = CH_AUX_OF(ABL);
    static_assert(NU % RING == 0, "unit count must be a multiple of the ring depth");
    ch_ex8 af[2][CH_NPL];
#pragma unroll 1
    for (int u0 = 0; u0 < NU; u0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int u = u0 + j;
            // request unit u + RING - 1 into the slot unit u - 1 has left
            const int un = u + RING - 1;
            if (CH_NOLOAD(ABL)) {
            } else if (un < NU) {
                const int an = (NACC == 2) ? (un & 1) : 0, ktn = (NACC == 2) ? (un >> 1) : un;
                ch_wload<RING, AUX, CH_NP_OF(ABL)>(R, (j + RING - 1) % RING, wrs, L, ((NACC == 2 && an) ? base1 : base0) + (unsigned)ktn * CH_WTILE);
            } else if (nx.nacc) {
                ch_wload<RING, AUX, CH_NP_OF(ABL)>(R, (j + RING - 1) % RING, wrs, L, nx.off(un - NU));
            }
            // (Measured and dropped, round 4: one extra dword load per unit that touches the 48 cache lines of the unit 4 / 8 / 12 ahead,
            //  to cover the memory-side latency inside a head step — 3.44 / 3.42-3.53 / 3.49 ms per 32-frame step against 3.38 without:
            //  the stream is bound by the CU's L2 -> L1 path, and a line prefetch moves every byte over that path twice.  Warming the
            //  memory-side cache from the gather reduction (k_gather_reduce `touch`) is what helps: 3.45 -> 3.38 ms.)
            // the request stays HERE, RING - 1 units ahead of its use: without the fence the machine scheduler sinks every load to just
            // before its MFMA (load, s_waitcnt vmcnt(0), mfma) to save registers — the whole point of the ring
            __builtin_amdgcn_sched_barrier(0);
            const int a = (NACC == 2) ? (u & 1) : 0, kt = (NACC == 2) ? (u >> 1) : u;
            if ((a == 0 || !SAMEA) && !(VKN_ABL_IS(ABL, 3) && u > 0)) {
                const ch_e* img = (a == 0) ? img0 : img1;
                ch_afrag(img, L, kt, 0, af[0]);
                ch_afrag(img, L, kt, 1, af[1]);
            }
            if (CH_NOMFMA(ABL)) {   // keep the loads and reads alive without the matrix pipe
#pragma unroll
                for (int f = 0; f < 2 * CH_NPL; ++f) asm volatile("" ::"v"(R.r[j][f]));
#pragma unroll
                for (int f = 0; f < CH_NPL; ++f) asm volatile("" ::"v"(af[0][f]), "v"(af[1][f]));
#if defined(VKN_DEBUG) && !defined(CH_H2)
            } else if (VKN_ABL_IS(ABL, 11)) {
                ch_mfma3(acc[a], R, j, 0, af[0]);
                ch_mfma3(acc[a], R, j, 1, af[1]);
#endif
            } else {
                ch_mfma(acc[a], R, j, 0, af[0]);
                ch_mfma(acc[a], R, j, 1, af[1]);
            }
        }
    }
}

__device__ __forceinline__ void ch_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// this lane's sixteen column constants of a [256]-float vector in LDS (columns wave * 32 + 8 q + 4 g + e)
__device__ __forceinline__ void ch_cols(const float* vec, const ChLane& L, float (&c)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(vec + L.wave * 32 + 8 * q + 4 * L.g);
#pragma unroll
        for (int e = 0; e < 4; ++e) c[4 * q + e] = t[e];
    }
}

// sum over the 256 columns of a row for NV per-lane partials: 16 partials per row (8 waves x 2 column halves) through LDS, summed in
// fixed order by every lane of the row.  One barrier.
template <int NV>
__device__ __forceinline__ void ch_rowsum(float (&p)[NV], float* Sbuf, const ChLane& L) {
#pragma unroll
    for (int v = 0; v < NV; ++v) Sbuf[v * 512 + (L.wave * 2 + L.g) * 32 + L.li] = p[v];
    CH_BAR();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += Sbuf[v * 512 + k * 32 + L.li];
        p[v] = s;
    }
}

// LayerNorm over the 256 columns of NV independent row sets held in registers (v[i][16]); weights / biases from LDS vectors.
// Two-pass (mean, centred variance), 1 / sqrtf, as the row epilogue of vkn_update.hip.  Two barriers; `sb` toggles the exchange buffer.
template <int NV>
__device__ __forceinline__ void ch_layernorm(float (&v)[NV][16], const float* const (&lw)[NV], const float* const (&lb)[NV], float eps,
                                             float* S, int& sb, const ChLane& L) {
    float p[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[i][r];
        p[i] = s;
    }
    ch_rowsum<NV>(p, S + sb * CH_SBUF, L);
    sb ^= 1;
    float mean[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        mean[i] = p[i] / 256.f;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = v[i][r] - mean[i];
            s += d * d;
        }
        p[i] = s;
    }
    ch_rowsum<NV>(p, S + sb * CH_SBUF, L);
    sb ^= 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float rstd = 1.0f / sqrtf(p[i] / 256.f + eps);
        float w[16], b[16];
        ch_cols(lw[i], L, w);
        ch_cols(lb[i], L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[i][r] = (v[i][r] - mean[i]) * rstd * w[r] + b[r];
    }
}

// registers (transposed tile) -> image: 8-byte packed stores, chunk (wave * 4 + q) of row li, half g
__device__ __forceinline__ void ch_img_write(ch_e* img, const float (&v)[16], const ChLane& L) {
#ifdef CH_H2
    {   // the images written here travel UNSCALED (LayerNorm outputs, ReLU / FFN hidden rows, branch inputs: O(1) for ordinary weights);
        // a value of 2^15 or more — large LayerNorm gains, a fine-tuned W1 — is one binade from fp16's inf: reported through the status
        // word instead of silently becoming NaN (the bf16x3 form, VKN_FLAG_CHAIN_BF16X3, has fp32 range).  NaN compares false: !(m < 2^15)
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
        bool bad = !(m < 32768.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) bad |= (v[r] != v[r]);
        if (bad && L.status) atomicOr(L.status, VKN_STATUS_RANGE);
    }
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float t[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        ch_ex4 o[CH_NPL];
        ch_split4(t, o);
        ch_e* d = img + L.li * CH_C + ((((L.wave << 2) + q) ^ L.li) & 31) * 8 + 4 * L.g;
#pragma unroll
        for (int i = 0; i < CH_NPL; ++i) *reinterpret_cast<ch_ex4*>(d + i * CH_PLANE) = o[i];
    }
}
#ifdef CH_H2
// the same for rows of unbounded magnitude (the updator's gate product): the row maximum over the workgroup's sixteen holders meets in
// `X` ([16][32] floats, one barrier), the row is written times the power of two that puts it at 2^9 .. 2^10; returns 1 / scale (the
// consumer GEMM's epilogue multiplies its accumulators by it — exact)
__device__ __forceinline__ float ch_img_write_rs(ch_e* img, const float (&v)[16], const ChLane& L, float* X) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
    X[(L.wave * 2 + L.g) * 32 + L.li] = m;
    CH_BAR();
#pragma unroll
    for (int k = 0; k < 16; ++k) m = fmaxf(m, X[k * 32 + L.li]);
    float inv;
    const float sc = ch_row_pow2(m, inv);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = v[r] * sc;
    ch_img_write(img, t, L);
    return inv;
}
#endif

// fp32 rows [M][ld] (32 rows from m0, 256 columns) -> image; rows >= M repeat the last row (never stored).  fp16 form: every row times
// its own power of two (row maximum over the sixteen lanes that hold the row), 1 / scale -> rs[row] (LDS, read behind the next barrier)
__device__ __forceinline__ void ch_img_load(ch_e* img, const float* __restrict__ src, int ld, int m0, int M, int tid, float* rs = nullptr) {
    const int row = tid >> 4, c4 = (tid & 15) << 2;
    const float* s = src + (size_t)min(m0 + row, M - 1) * ld + c4;
    f32x4 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const f32x4*>(s + 64 * j);
#ifdef CH_H2
    {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(t[j][e]));
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        float inv;
        const float sc = ch_row_pow2(m, inv);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= sc;
        if ((tid & 15) == 0) rs[row] = inv;
    }
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = c4 + 64 * j;
        const float tv[4] = {t[j][0], t[j][1], t[j][2], t[j][3]};
        ch_ex4 o[CH_NPL];
        ch_split4(tv, o);
        ch_e* d = img + row * CH_C + ((((col >> 3) ^ row) & 31) << 3) + (col & 7);
#pragma unroll
        for (int i = 0; i < CH_NPL; ++i) *reinterpret_cast<ch_ex4*>(d + i * CH_PLANE) = o[i];
    }
}
// fp16 form: acc *= u (u = 1 / (weight image scale x activation row scale), powers of two: exact); bf16 form: nothing
#ifdef CH_H2
#define CH_UNSCALE(A, U)                                     \
    do {                                                     \
        const float u_ = (U);                                \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)(A)[r_] *= u_; \
    } while (0)
#else
#define CH_UNSCALE(A, U) \
    do {                 \
    } while (0)
#endif

// registers (transposed tile) -> fp32 rows [M][ld] at column tile offset col0: four float4 stores per lane
__device__ __forceinline__ void ch_store_rows(float* __restrict__ dst, int ld, int col0, int row, bool ok, const float (&v)[16],
                                              const ChLane& L) {
    if (!ok) return;
    float* d = dst + (size_t)row * ld + col0 + L.wave * 32 + 4 * L.g;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(d + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

// a lane's sixteen values parked in LDS across a GEMM (register pressure): private slots [q][thread] float4, no barrier needed
__device__ __forceinline__ void ch_park(float* P, const float (&v)[16], int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(P + (q * CH_THREADS + tid) * 4) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
__device__ __forceinline__ void ch_unpark(const float* P, float (&v)[16], int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(P + (q * CH_THREADS + tid) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
    }
}

// the kernel's constant vectors -> LDS: ONE contiguous, 16-byte aligned block packed at prepare time (k_chain_pack), copied with all
// loads in flight at once (a table of separate vectors cost one global round trip per vector: ~7 us per launch)
template <int NF>  // floats, multiple of 4
__device__ __forceinline__ void ch_stage_consts(float* cst, const float* __restrict__ src, int tid) {
    constexpr int NV = NF / 4, PER = (NV + CH_THREADS - 1) / CH_THREADS;
    f32x4 t[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid + i * CH_THREADS;
        t[i] = *reinterpret_cast<const f32x4*>(src + 4 * min(k, NV - 1));
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = tid + i * CH_THREADS;
        if (k < NV) *reinterpret_cast<f32x4*>(cst + 4 * k) = t[i];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ kernel A: updator + in_proj
// LDS constant map of k_chain_a (floats)
enum {
    CA_DYN_B = 0,      // [512] dynamic_layer bias (scaled per row by `rowscale` when the folded feat_transform bias is used)
    CA_DYN_B2 = 512,   // [512] second, unscaled bias (dyn_b beside the folded W_dyn.b_ft) or zeros
    CA_NO_W = 1024, CA_NO_B = 1280,    // norm_out
    CA_INP_B = 1536,   // [512]
    CA_INO_W = 2048, CA_INO_B = 2304,  // input_norm_out
    CA_IG_B = 2560, CA_INI_W = 2816, CA_INI_B = 3072,   // input_gate bias, input_norm_in
    CA_UG_B = 3328, CA_NI_W = 3584, CA_NI_B = 3840,     // update_gate bias, norm_in
    CA_FC_B = 4096, CA_FCN_W = 4352, CA_FCN_B = 4608,   // fc_layer bias, fc_norm
    CA_IN_B = 4864,    // [768] attention in_proj bias
    CA_SC = 5632,      // [8] fp16 form: 1 / scale of the weight images dyn (this block's mode), inp, ig, ug, fc, in_proj (1 in the bf16 form)
    CA_TOTAL = 5640
};

struct ChainAArgs {
    const float* a0;        // [M][256] update feature: the raw gather (composite dynamic weights) or x_feat
    const float* obj_in;    // [M][256] incoming kernels
    const float* rowscale;  // [M] pixel counts (scale of CA_DYN_B) or NULL
    const void* wbase;      // prepared weight buffer
    unsigned wbytes;
    unsigned off_dyn, off_inp, off_ig, off_ug, off_fc, off_in;  // tile images (byte offsets in wbase)
    float eps;
    int M;
    float* obj1;  // [M][256] out: updated kernels (residual of the attention block)
    float* qkv;   // [M][768] out: packed q | k | v
    const float* consts;   // [CA_TOTAL] packed block (vkn_chain_pack_consts)
    int* status;
};

template <int ABL>
__global__ __launch_bounds__(CH_THREADS) void k_chain_a(const ChainAArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_ca[];
    ch_e* IMG0 = reinterpret_cast<ch_e*>(smem_ca);
    ch_e* IMG1 = IMG0 + CH_IMG;
    float* S = reinterpret_cast<float*>(IMG1 + CH_IMG);  // [2][CH_SBUF]
    float* CST = S + 2 * CH_SBUF;
    float* PARK = CST + CA_TOTAL;                        // [4][512] float4: parameters_in, later input_out, parked across a GEMM
#ifdef CH_H2
    float* RS = PARK + 16 * CH_THREADS;                  // [2][32] 1 / row scale of the two loaded images; [16][32] row-maximum exchange
#endif

    const int tid = threadIdx.x;
    ChLane L_ = ch_lane(tid);
    L_.status = A.status;
    const ChLane L = L_;
    const int m0 = blockIdx.x * CH_ROWS, M = A.M;
    const int row = m0 + L.li;
    const bool row_ok = row < M;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.wbase), 0, (int)A.wbytes, 0x00020000);

    constexpr int RING = CH_RING_OF(ABL);
    ChRing<RING> R;
    {   // the first RING - 1 units of the dynamic_layer GEMM (unit u = column tile u & 1 of K-tile u >> 1)
        const ChNext first{A.off_dyn, A.off_dyn + 8u * CH_WTILE, 2};
#pragma unroll
        for (int j = 0; j < RING - 1; ++j) ch_wload<RING, CH_AUX_OF(ABL), CH_NP_OF(ABL)>(R, j, wrs, L, first.off(j));
        if (CH_NOLOAD(ABL)) ch_wload<RING>(R, RING - 1, wrs, L, A.off_dyn);   // (ablation: the ring is never refilled; defined contents)
    }
    ch_stage_consts<CA_TOTAL>(CST, A.consts, tid);
#ifdef CH_H2
    ch_img_load(IMG0, A.a0, CH_C, m0, M, tid, RS);
    ch_img_load(IMG1, A.obj_in, CH_C, m0, M, tid, RS + 32);
#else
    ch_img_load(IMG0, A.a0, CH_C, m0, M, tid);
    ch_img_load(IMG1, A.obj_in, CH_C, m0, M, tid);
#endif
    const float bs = A.rowscale ? A.rowscale[min(row, M - 1)] : 1.f;
    CH_BAR();

    int sb = 0;
    f32x16 acc[2];
    float pout[1][16];
    // ---- dynamic_layer(update feature): parameters_in | LN(parameters_out)                        knet/kernel_updator.py:59-62, :79
    ch_zero(acc[0]);
    ch_zero(acc[1]);
    ch_gemm<2, true, ABL>(acc, IMG0, IMG0, A.off_dyn, A.off_dyn + 8u * CH_WTILE, ChNext{A.off_inp, A.off_inp + 8u * CH_WTILE, 2}, R, wrs, L);
    CH_UNSCALE(acc[0], CST[CA_SC + 0] * RS[L.li]);
    CH_UNSCALE(acc[1], CST[CA_SC + 0] * RS[L.li]);
    {
        float b[16], b2[16], pin[16];
        ch_cols(CST + CA_DYN_B, L, b);
        ch_cols(CST + CA_DYN_B2, L, b2);
#pragma unroll
        for (int r = 0; r < 16; ++r) pin[r] = acc[0][r] + b[r] * bs + b2[r];
        ch_park(PARK, pin, tid);
        ch_cols(CST + CA_DYN_B + 256, L, b);
        ch_cols(CST + CA_DYN_B2 + 256, L, b2);
#pragma unroll
        for (int r = 0; r < 16; ++r) pout[0][r] = acc[1][r] + b[r] * bs + b2[r];
        const float* const lw[1] = {CST + CA_NO_W};
        const float* const lb[1] = {CST + CA_NO_B};
        ch_layernorm<1>(pout, lw, lb, A.eps, S, sb, L);
    }
    // ---- input_layer(kernels): input_in | LN(input_out); gate = input_in * parameters_in -> image             :65-70, :80
    ch_zero(acc[0]);
    ch_zero(acc[1]);
    ch_gemm<2, true, ABL>(acc, IMG1, IMG1, A.off_inp, A.off_inp + 8u * CH_WTILE, ChNext{A.off_ig, A.off_ug, 2}, R, wrs, L);
    CH_UNSCALE(acc[0], CST[CA_SC + 1] * RS[32 + L.li]);
    CH_UNSCALE(acc[1], CST[CA_SC + 1] * RS[32 + L.li]);
#ifdef CH_H2
    float gate_inv;
#endif
    {
        float b[16], gate[16], iout[1][16];
        ch_cols(CST + CA_INP_B, L, b);
        ch_unpark(PARK, gate, tid);
#pragma unroll
        for (int r = 0; r < 16; ++r) gate[r] = (acc[0][r] + b[r]) * gate[r];
#ifdef CH_H2
        gate_inv = ch_img_write_rs(IMG0, gate, L, RS + 64);   // (an unnormalised product: its rows are scaled one by one)
#else
        ch_img_write(IMG0, gate, L);   // IMG0 was last read by the first GEMM: every wave is past that epilogue's barriers
#endif
        ch_cols(CST + CA_INP_B + 256, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) iout[0][r] = acc[1][r] + b[r];
        const float* const lw[1] = {CST + CA_INO_W};
        const float* const lb[1] = {CST + CA_INO_B};
        ch_layernorm<1>(iout, lw, lb, A.eps, S, sb, L);   // (its barriers publish the gate image)
        ch_park(PARK, iout[0], tid);
    }
    // ---- input_gate / update_gate = sigmoid(LN(linear(gate))); features = update_gate * param_out + input_gate * input_out   :72-88
    ch_zero(acc[0]);
    ch_zero(acc[1]);
    ch_gemm<2, true, ABL>(acc, IMG0, IMG0, A.off_ig, A.off_ug, ChNext{A.off_fc, 0u, 1}, R, wrs, L);
#ifdef CH_H2
    CH_UNSCALE(acc[0], CST[CA_SC + 2] * gate_inv);
    CH_UNSCALE(acc[1], CST[CA_SC + 3] * gate_inv);
#endif
    {
        float gt[2][16], b[16];
        ch_cols(CST + CA_IG_B, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) gt[0][r] = acc[0][r] + b[r];
        ch_cols(CST + CA_UG_B, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) gt[1][r] = acc[1][r] + b[r];
        const float* const lw[2] = {CST + CA_INI_W, CST + CA_NI_W};
        const float* const lb[2] = {CST + CA_INI_B, CST + CA_NI_B};
        ch_layernorm<2>(gt, lw, lb, A.eps, S, sb, L);
        float f[16];
        ch_unpark(PARK, f, tid);      // input_out
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ig = 1.0f / (1.0f + expf(-gt[0][r]));
            const float ug = 1.0f / (1.0f + expf(-gt[1][r]));
            f[r] = ug * pout[0][r] + ig * f[r];
        }
        ch_img_write(IMG1, f, L);   // IMG1 was last read by the input_layer GEMM
        CH_BAR();
    }
    // ---- fc_layer + fc_norm + ReLU -> updated kernels (obj1)                                                      :90-92
    f32x16 acc1[1];
    ch_zero(acc1[0]);
    ch_gemm<1, true, ABL>(acc1, IMG1, IMG1, A.off_fc, 0u, ChNext{A.off_in, 0u, 1}, R, wrs, L);
    CH_UNSCALE(acc1[0], CST[CA_SC + 4]);
    {
        float o[1][16], b[16];
        ch_cols(CST + CA_FC_B, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[0][r] = acc1[0][r] + b[r];
        const float* const lw[1] = {CST + CA_FCN_W};
        const float* const lb[1] = {CST + CA_FCN_B};
        ch_layernorm<1>(o, lw, lb, A.eps, S, sb, L);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[0][r] = fmaxf(o[0][r], 0.f);
        ch_store_rows(A.obj1, CH_C, 0, row, row_ok, o[0], L);
        ch_img_write(IMG0, o[0], L);   // IMG0 was last read by the gate GEMM
        CH_BAR();
    }
    // ---- attention in_proj: q | k | v = obj1 . W_in^T + b_in                                     knet/det/kernel_update_head.py:206
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        asm volatile("" ::: "memory");   // the image reads are loop invariant: keep hipcc from hoisting all 48 fragments (192 VGPRs) out of the loop
        ch_zero(acc1[0]);
        const ChNext nx = (t < 2) ? ChNext{A.off_in + (unsigned)(t + 1) * 8u * CH_WTILE, 0u, 1} : ChNext{0u, 0u, 0};
        ch_gemm<1, true, ABL>(acc1, IMG0, IMG0, A.off_in + (unsigned)t * 8u * CH_WTILE, 0u, nx, R, wrs, L);
        CH_UNSCALE(acc1[0], CST[CA_SC + 5]);
        float o[16], b[16];
        ch_cols(CST + CA_IN_B + 256 * t, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc1[0][r] + b[r];
        ch_store_rows(A.qkv, 3 * CH_C, 256 * t, row, row_ok, o, L);
    }
}

// ------------------------------------------------------------------------------------------------ kernel C: out_proj .. decode kernels
enum {
    CC_OUT_B = 0, CC_AN_W = 256, CC_AN_B = 512,       // attention out_proj bias, attention_norm
    CC_B2 = 768, CC_FN_W = 1024, CC_FN_B = 1280,      // ffn second bias, ffn_norm
    CC_CLN_W = 1536, CC_CLN_B = 1792,                 // cls_fcs LayerNorm
    CC_MLN_W = 2048, CC_MLN_B = 2304,                 // mask_fcs LayerNorm
    CC_DVEC = 2560,                                   // W_fm^T . b_ft (decode-bias dot vector)
    CC_CLS_B = 2816,                                  // fc_cls bias, zero padded to 256
    CC_DEC_B = 3072,                                  // bias of the folded decode kernels
    CC_B1 = 3328,                                     // [ff <= 2048] ffn first bias
    CC_SC = 3328 + 2048,                              // [8] fp16 form: 1 / scale of the weight images out, ffn1, ffn2, cls_fc, mask_fc, fc_cls, dec
    CC_TOTAL = 3328 + 2048 + 8
};

struct ChainCArgs {
    const float* ao;     // [M][256] attention output (before out_proj)
    const float* obj1;   // [M][256] residual of the attention block
    const void* wbase;
    unsigned wbytes;
    unsigned off_out, off_ffn1, off_ffn2, off_clsfc, off_maskfc, off_fccls, off_dec;
    int nchunks;         // ff / 256
    int has_cls;         // fc_cls is computed
    int cls_sigmoid;
    int ncls;
    float eps;
    int M;
    const float* kb0;    // device scalar b_fm . b_ft (added to the decode bias) or NULL
    float* obj_out;      // [M][256] the stage's output kernels
    float* cls_out;      // [M][ncls] or NULL
    float* kb_out;       // [M] folded decode bias
    _Float16* plane_hi;  // f16 split planes [B][NPT][256] of the folded decode kernels, or NULL ->
    _Float16* plane_lo;
    float* kern_out;     // ... fp32 [M][256]
    int rows_per_frame, NPT;
    const float* consts;   // [CC_TOTAL] packed block
    int* status;
};

template <int ABL>
__global__ __launch_bounds__(CH_THREADS) void k_chain_c(const ChainCArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_cc[];
    ch_e* IMG0 = reinterpret_cast<ch_e*>(smem_cc);
    ch_e* HID = IMG0 + CH_IMG;
    float* S = reinterpret_cast<float*>(HID + CH_IMG);
    float* CST = S + 2 * CH_SBUF;
#ifdef CH_H2
    float* RS = CST + CC_TOTAL;   // [32] 1 / row scale of the attention-output image
#endif

    const int tid = threadIdx.x;
    ChLane L_ = ch_lane(tid);
    L_.status = A.status;
    const ChLane L = L_;
    const int m0 = blockIdx.x * CH_ROWS, M = A.M;
    const int row = m0 + L.li;
    const bool row_ok = row < M;
    const int rowc = min(row, M - 1);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.wbase), 0, (int)A.wbytes, 0x00020000);

    constexpr int RING = CH_RING_OF(ABL);
    ChRing<RING> R;
    {
        const ChNext first{A.off_out, 0u, 1};
#pragma unroll
        for (int j = 0; j < RING - 1; ++j) ch_wload<RING, CH_AUX_OF(ABL), CH_NP_OF(ABL)>(R, j, wrs, L, first.off(j));
        if (CH_NOLOAD(ABL)) ch_wload<RING>(R, RING - 1, wrs, L, A.off_out);
    }
    ch_stage_consts<CC_TOTAL>(CST, A.consts, tid);
#ifdef CH_H2
    ch_img_load(IMG0, A.ao, CH_C, m0, M, tid, RS);
#else
    ch_img_load(IMG0, A.ao, CH_C, m0, M, tid);
#endif
    float obj[1][16];   // residual rows in the transposed-tile layout
    {
        const float* s = A.obj1 + (size_t)rowc * CH_C + L.wave * 32 + 4 * L.g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(s + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) obj[0][4 * q + e] = t[e];
        }
    }
    CH_BAR();

    int sb = 0;
    f32x16 acc1[1];
    // ---- attention out_proj + identity + attention_norm                                           knet/det/kernel_update_head.py:206-208
    ch_zero(acc1[0]);
    ch_gemm<1, true, ABL>(acc1, IMG0, IMG0, A.off_out, 0u, ChNext{A.off_ffn1, 0u, 1}, R, wrs, L);
    CH_UNSCALE(acc1[0], CST[CC_SC + 0] * RS[L.li]);
    {
        float b[16];
        ch_cols(CST + CC_OUT_B, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) obj[0][r] = acc1[0][r] + b[r] + obj[0][r];
        const float* const lw[1] = {CST + CC_AN_W};
        const float* const lb[1] = {CST + CC_AN_B};
        ch_layernorm<1>(obj, lw, lb, A.eps, S, sb, L);
        ch_img_write(IMG0, obj[0], L);   // every wave has finished the out_proj reads of IMG0 (the LayerNorm barriers)
        CH_BAR();
    }
    // ---- FFN: obj + W2 relu(W1 obj + b1) + b2, hidden chunks of 256 in sequence; the hidden activations live in ONE image     :214-215
    f32x16 acc2[1];
    ch_zero(acc2[0]);
#pragma unroll 1
    for (int c = 0; c < A.nchunks; ++c) {
        ch_zero(acc1[0]);
        ch_gemm<1, true, ABL>(acc1, IMG0, IMG0, A.off_ffn1 + (unsigned)c * 8u * CH_WTILE, 0u,
                         ChNext{A.off_ffn2 + (unsigned)c * 8u * CH_WTILE, 0u, 1}, R, wrs, L);
        CH_UNSCALE(acc1[0], CST[CC_SC + 1]);
        float h[16], b[16];
        ch_cols(CST + CC_B1 + 256 * c, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = fmaxf(acc1[0][r] + b[r], 0.f);
        CH_BAR();                      // the previous chunk's second GEMM has finished reading the hidden image (all waves)
        ch_img_write(HID, h, L);
        CH_BAR();
        const bool lastc = (c + 1 == A.nchunks);
        const ChNext nx = lastc ? ChNext{A.off_clsfc, A.off_maskfc, 2} : ChNext{A.off_ffn1 + (unsigned)(c + 1) * 8u * CH_WTILE, 0u, 1};
        ch_gemm<1, true, ABL>(acc2, HID, HID, A.off_ffn2 + (unsigned)c * 8u * CH_WTILE, 0u, nx, R, wrs, L);
    }
    CH_UNSCALE(acc2[0], CST[CC_SC + 2]);   // (one scale per weight image: the hidden chunks' partial products share it)
    {
        float b[16];
        ch_cols(CST + CC_B2, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) obj[0][r] = obj[0][r] + acc2[0][r] + b[r];
        const float* const lw[1] = {CST + CC_FN_W};
        const float* const lb[1] = {CST + CC_FN_B};
        ch_layernorm<1>(obj, lw, lb, A.eps, S, sb, L);
        ch_store_rows(A.obj_out, CH_C, 0, row, row_ok, obj[0], L);
        ch_img_write(IMG0, obj[0], L);   // IMG0's last reader was the last chunk's first GEMM (barriers since)
        CH_BAR();
    }
    // ---- cls_fcs[0] / mask_fcs[0]: Linear (no bias) + LN + ReLU, one pass over the obj image                      :217-226
    f32x16 acc[2];
    ch_zero(acc[0]);
    ch_zero(acc[1]);
    const ChNext nfin = A.has_cls ? ChNext{A.off_fccls, A.off_dec, 2} : ChNext{A.off_dec, 0u, 1};
    ch_gemm<2, true, ABL>(acc, IMG0, IMG0, A.off_clsfc, A.off_maskfc, nfin, R, wrs, L);
    CH_UNSCALE(acc[0], CST[CC_SC + 3]);
    CH_UNSCALE(acc[1], CST[CC_SC + 4]);
    {
        float t[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t[0][r] = acc[0][r];
            t[1][r] = acc[1][r];
        }
        const float* const lw[2] = {CST + CC_CLN_W, CST + CC_MLN_W};
        const float* const lb[2] = {CST + CC_CLN_B, CST + CC_MLN_B};
        ch_layernorm<2>(t, lw, lb, A.eps, S, sb, L);
        float dv[16], kd[1];
        ch_cols(CST + CC_DVEC, L, dv);
        kd[0] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t[0][r] = fmaxf(t[0][r], 0.f);
            t[1][r] = fmaxf(t[1][r], 0.f);
            kd[0] += t[1][r] * dv[r];
        }
        ch_img_write(IMG0, t[0], L);   // cls branch input of fc_cls
        ch_img_write(HID, t[1], L);    // mask branch input of the folded fc_mask
        ch_rowsum<1>(kd, S + sb * CH_SBUF, L);   // folded decode bias kb = mask_feat . b_ft; its barrier publishes both images
        sb ^= 1;
        if (row_ok && L.wave == 0 && L.g == 0 && A.kb_out) A.kb_out[row] = kd[0] + (A.kb0 ? *A.kb0 : 0.f);
    }
    // ---- fc_cls (+ sigmoid on the last stage) and the folded decode kernels Kf = fc_mask(.) . W_ft                 :221, :227, :247
    ch_zero(acc[0]);
    ch_zero(acc[1]);
    if (A.has_cls) {
        ch_gemm<2, false, ABL>(acc, IMG0, HID, A.off_fccls, A.off_dec, ChNext{0u, 0u, 0}, R, wrs, L);
        CH_UNSCALE(acc[0], CST[CC_SC + 5]);
        if (L.wave * 32 < A.ncls && row_ok && A.cls_out) {
            float b[16];
            ch_cols(CST + CC_CLS_B, L, b);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = L.wave * 32 + 8 * (r >> 2) + 4 * L.g + (r & 3);
                float v = acc[0][r] + b[r];
                if (A.cls_sigmoid) v = 1.0f / (1.0f + expf(-v));
                if (col < A.ncls) A.cls_out[(size_t)row * A.ncls + col] = v;
            }
        }
    } else {
        f32x16 accd[1];
        ch_zero(accd[0]);
        ch_gemm<1, true, ABL>(accd, HID, HID, A.off_dec, 0u, ChNext{0u, 0u, 0}, R, wrs, L);
        acc[1] = accd[0];
    }
    CH_UNSCALE(acc[1], CST[CC_SC + 6]);
    {
        float k[16], b[16];
        ch_cols(CST + CC_DEC_B, L, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) k[r] = acc[1][r] + b[r];
        if (A.plane_hi) {
            if (row_ok) {
                const int fb = row / A.rows_per_frame, n = row - fb * A.rows_per_frame;
                const size_t base = ((size_t)fb * A.NPT + n) * CH_C + L.wave * 32 + 4 * L.g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 hh, ll;
                        vkn_split_f16(k[4 * q + e], hh, ll);
                        h[e] = hh;
                        l[e] = ll;
                    }
                    *reinterpret_cast<half4*>(A.plane_hi + base + 8 * q) = h;
                    *reinterpret_cast<half4*>(A.plane_lo + base + 8 * q) = l;
                }
            }
        } else {
            ch_store_rows(A.kern_out, CH_C, 0, row, row_ok, k, L);
        }
    }
}

#ifndef CH_H2
// ------------------------------------------------------------------------------------------------ one GEMM per launch on the same engine
// k_gemm_t3: out = epi((A (.* A2) (+ A3 .* A4)) . W^T) for K == 256 KC (KC <= 3: the backward GEMMs of the training chain contract over 512 / 768) with the row epilogue of k_gemm_s3 (vkn_update.hip: VknEpi) — the
// launch-per-GEMM chain of few-row calls (frame-by-frame video inference: 117 rows), the link blocks and the stand-alone linear entry
// points.  Same tile (32 rows x 256 columns per workgroup, wave = column block), same six products per operand pair, but the K loop is
// the chain kernels': the A tile is split ONCE into a resident LDS image (K = 256 whole), the weight fragments go straight from
// global memory into the consumer wave's register ring, no barrier and no LDS-DMA inside the loop, the epilogue runs in registers
// (LayerNorm through the 16-partial exchange).  k_gemm_s3's loop cost 0.87 us per K-tile with a barrier each (7 us of a 13.8 us
// launch at 117 rows); this one streams at ~0.5 us per tile.  Up to two problems per launch (blockIdx.z), as k_gemm_s3.
template <int ABL, int KC>
__global__ __launch_bounds__(CH_THREADS) void k_gemm_t3(const VknGemmProb p0, const VknGemmProb p1, int nprob, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem_t3[];
    ch_e* IMG = reinterpret_cast<ch_e*>(smem_t3);
    float* S = reinterpret_cast<float*>(IMG + KC * CH_IMG);   // KC images of 256 K-columns each (K = 256 KC <= 768: 144 KB + S)
    const bool second = (nprob > 1) && (blockIdx.z == 1);
    const VknGemmProb& P = second ? p1 : p0;
    const VknEpi& E = P.epi;
    const int Nout = P.Nout;
    const int m0 = blockIdx.y * CH_ROWS, n0 = blockIdx.x * 256;
    if (n0 >= Nout) return;   // grouped launch: the grid is sized for the wider problem (uniform exit before any barrier)

    const int tid = threadIdx.x;
    const ChLane L = ch_lane(tid);
    const int ntiles = (Nout + 255) / 256;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.Wsplit), 0, (int)((unsigned)ntiles * (8u * KC) * CH_WTILE), 0x00020000);
    const unsigned wbase = (unsigned)blockIdx.x * (8u * KC) * CH_WTILE;
    constexpr int RING = CH_RING_OF(ABL);
    ChRing<RING> R;
#pragma unroll
    for (int j = 0; j < RING - 1; ++j) ch_wload<RING, CH_AUX_OF(ABL)>(R, j, wrs, L, wbase + (unsigned)j * CH_WTILE);

    // ---- the A tile -> bf16x3 images (optional elementwise prologue: A .* A2 + A3 .* A4, the updator's gate / mix)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const int row = tid >> 4, c4 = (tid & 15) << 2;
        const size_t ro = (size_t)min(m0 + row, M - 1) * P.lda + c4 + 256 * kc;
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const f32x4*>(P.A + ro + 64 * j);
        if (P.A2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] *= *reinterpret_cast<const f32x4*>(P.A2 + ro + 64 * j);
        }
        if (P.A3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                t[j] += *reinterpret_cast<const f32x4*>(P.A3 + ro + 64 * j) * *reinterpret_cast<const f32x4*>(P.A4 + ro + 64 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c4 + 64 * j;
            const float tv[4] = {t[j][0], t[j][1], t[j][2], t[j][3]};
            ch_ex4 o[CH_NPL];
            ch_split4(tv, o);
            ch_e* d = IMG + kc * CH_IMG + row * CH_C + ((((col >> 3) ^ row) & 31) << 3) + (col & 7);
#pragma unroll
            for (int i = 0; i < CH_NPL; ++i) *reinterpret_cast<ch_ex4*>(d + i * CH_PLANE) = o[i];
        }
    }
    // ---- the lane's sixteen columns and their epilogue constants (requested before the K loop: their round trips ride under it)
    const int row = m0 + L.li;
    const bool row_ok = row < M;
    const int rowc = min(row, M - 1);
    const int ncols = min(256, Nout - n0);
    const bool do_ln = (E.ln_w != nullptr) && (n0 >= E.ln_from_col);
    float bias[16], lnw[16], lnb[16], rv[16];
    bool okc[16];
    const float bscale = (E.bias && E.rowscale) ? E.rowscale[rowc] : 1.f;
    // a full, 16-byte aligned tile (every chain GEMM but fc_cls): four float4 loads per vector instead of sixteen dword loads
    const bool vec4 = ncols == 256 && ((reinterpret_cast<uintptr_t>(E.bias) | reinterpret_cast<uintptr_t>(E.bias2) | reinterpret_cast<uintptr_t>(E.ln_w) |
                                        reinterpret_cast<uintptr_t>(E.ln_b) | reinterpret_cast<uintptr_t>(E.resid)) & 15) == 0 && (E.ldr & 3) == 0;
    if (vec4) {
        const int cb = n0 + L.wave * 32 + 4 * L.g;
        auto ld4 = [&](const float* p, float dflt, float (&o)[16]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = p ? *reinterpret_cast<const f32x4*>(p + 8 * q) : f32x4{dflt, dflt, dflt, dflt};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * q + e] = t[e];
            }
        };
        float b2[16];
        ld4(E.bias ? E.bias + cb : nullptr, 0.f, bias);
        ld4(E.bias2 ? E.bias2 + cb : nullptr, 0.f, b2);
        ld4(do_ln ? E.ln_w + (cb - E.ln_from_col) : nullptr, 1.f, lnw);
        ld4(do_ln ? E.ln_b + (cb - E.ln_from_col) : nullptr, 0.f, lnb);
        ld4(E.resid ? E.resid + (size_t)rowc * E.ldr + cb : nullptr, 0.f, rv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            okc[r] = true;
            bias[r] = bias[r] * bscale + b2[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = L.wave * 32 + 8 * (r >> 2) + 4 * L.g + (r & 3);   // column inside the tile
            okc[r] = cl < ncols;
            const int c = n0 + min(cl, ncols - 1);
            float b = E.bias ? E.bias[c] * bscale : 0.f;
            if (E.bias2) b += E.bias2[c];
            bias[r] = b;
            lnw[r] = do_ln ? E.ln_w[c - E.ln_from_col] : 1.f;
            lnb[r] = do_ln ? E.ln_b[c - E.ln_from_col] : 0.f;
            rv[r] = E.resid ? E.resid[(size_t)rowc * E.ldr + c] : 0.f;
        }
    }
    CH_BAR();

    f32x16 acc[1];
    ch_zero(acc[0]);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)   // the weight stream runs on across the K chunks
        ch_gemm<1, true, ABL>(acc, IMG + kc * CH_IMG, IMG + kc * CH_IMG, wbase + (unsigned)kc * 8u * CH_WTILE, 0u,
                              ChNext{kc + 1 < KC ? wbase + (unsigned)(kc + 1) * 8u * CH_WTILE : 0u, 0u, kc + 1 < KC ? 1 : 0}, R, wrs, L);

    float v[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[0][r] = okc[r] ? (acc[0][r] + bias[r] + rv[r]) : 0.f;
    if (do_ln) {   // LayerNorm over the tile's ncols columns (two-pass), as vkn_row_epilogue
        float pr[1];
        pr[0] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[0] += v[0][r];
        ch_rowsum<1>(pr, S, L);
        const float mean = pr[0] / (float)ncols;
        pr[0] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = okc[r] ? (v[0][r] - mean) : 0.f;
            pr[0] += dlt * dlt;
        }
        ch_rowsum<1>(pr, S + CH_SBUF, L);
        const float rstd = 1.0f / sqrtf(pr[0] / (float)ncols + E.eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[0][r] = (v[0][r] - mean) * rstd * lnw[r] + lnb[r];
    }
    if (E.act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[0][r] = fmaxf(v[0][r], 0.f);
    } else if (E.act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[0][r] = 1.0f / (1.0f + expf(-v[0][r]));
    }
    if (E.out && row_ok) {
        float* o = E.out + (size_t)row * E.ldo + n0 + L.wave * 32 + 4 * L.g;
        const bool vec = ((E.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(E.out) & 15) == 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (vec && okc[4 * q + 3]) {
                *reinterpret_cast<f32x4*>(o + 8 * q) = f32x4{v[0][4 * q], v[0][4 * q + 1], v[0][4 * q + 2], v[0][4 * q + 3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (okc[4 * q + e]) o[8 * q + e] = v[0][4 * q + e];
            }
        }
    }
    if (E.dot_vec) {   // (uniform) side output: dot_out[row] = sum_col v * dot_vec[col] (+ *dot_bias) — the folded decode bias
        float pr[1];
        pr[0] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = L.wave * 32 + 8 * (r >> 2) + 4 * L.g + (r & 3);
            pr[0] += okc[r] ? v[0][r] * E.dot_vec[n0 + cl] : 0.f;
        }
        CH_BAR();                       // (the exchange buffer may still be read by a slower wave of the LayerNorm above)
        ch_rowsum<1>(pr, S, L);
        if (row_ok && L.wave == 0 && L.g == 0) E.dot_out[row] = pr[0] + (E.dot_bias ? *E.dot_bias : 0.f);
    }
    if (E.plane_hi && row_ok) {
        const int fb = row / E.rows_per_frame, n = row - fb * E.rows_per_frame;
        const size_t base = ((size_t)fb * E.NPT + n) * E.ldo + n0 + L.wave * 32 + 4 * L.g;
        const bool vec = ((E.ldo & 3) == 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 hh, ll;
                vkn_split_f16(v[0][4 * q + e], hh, ll);
                h[e] = hh;
                l[e] = ll;
            }
            if (vec && okc[4 * q + 3]) {
                *reinterpret_cast<half4*>(E.plane_hi + base + 8 * q) = h;
                *reinterpret_cast<half4*>(E.plane_lo + base + 8 * q) = l;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (okc[4 * q + e]) {
                        E.plane_hi[base + 8 * q + e] = h[e];
                        E.plane_lo[base + 8 * q + e] = l[e];
                    }
            }
        }
    }
}

// K in {256, 512, 768}, pre-split weights, no split-K: the problems of one launch (same M, same K).  Returns VKN_E_SHAPE when the kernel
// does not apply (the caller then takes k_gemm_s3).
int vkn_launch_gemm_t3(const VknGemmProb* probs, int nprob, int M, int K, hipStream_t stream) {
    if (nprob < 1 || nprob > 2 || M <= 0) return VKN_E_ARG;
    if (K != 256 && K != 512 && K != 768) return VKN_E_SHAPE;
    const int kc = K / 256;
    int nmax = 0;
    for (int i = 0; i < nprob; ++i) {
        const VknGemmProb& p = probs[i];
        if (!p.Wsplit || (p.lda & 3) || (reinterpret_cast<uintptr_t>(p.A) & 15) || (p.A2 && (reinterpret_cast<uintptr_t>(p.A2) & 15)) ||
            (p.A3 && ((reinterpret_cast<uintptr_t>(p.A3) & 15) || !p.A4 || (reinterpret_cast<uintptr_t>(p.A4) & 15))))
            return VKN_E_SHAPE;
        if ((size_t)((p.Nout + 255) / 256) * 8 * kc * CH_WTILE >= (1ull << 31)) return VKN_E_SHAPE;
        nmax = p.Nout > nmax ? p.Nout : nmax;
    }
    const size_t lds = (size_t)kc * CH_IMG * sizeof(ch_e) + (size_t)2 * CH_SBUF * sizeof(float);
    dim3 grid((nmax + 255) / 256, (M + CH_ROWS - 1) / CH_ROWS, nprob);
#define T3_LAUNCH(KCV)                                                                                                                  \
    do {                                                                                                                                \
        VKN_ALLOW_FULL_LDS((k_gemm_t3<0, KCV>));                                                                                        \
        hipLaunchKernelGGL((k_gemm_t3<0, KCV>), grid, dim3(CH_THREADS), lds, stream, probs[0], probs[nprob > 1 ? 1 : 0], nprob, M);     \
    } while (0)
    if (kc == 1) T3_LAUNCH(1);
    else if (kc == 2) T3_LAUNCH(2);
    else T3_LAUNCH(3);
#undef T3_LAUNCH
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// ------------------------------------------------------------------------------------------------ host launchers
// constant blocks: packed once per weight update into the prepared buffer (vkn_prepare_stage_f32)
#define CH_PACK_MAX 96
struct ChPackTab {
    const float* src[CH_PACK_MAX];
    int dst[CH_PACK_MAX], n[CH_PACK_MAX];
    float fill[CH_PACK_MAX];
    int count;
};
__global__ __launch_bounds__(256) void k_chain_pack(const ChPackTab T, float* __restrict__ out) {
    const int i = blockIdx.x;
    const float* s = T.src[i];
    for (int k = threadIdx.x; k < T.n[i]; k += 256) out[T.dst[i] + k] = s ? s[k] : T.fill[i];
}
static void ch_tab_add(ChPackTab& T, const float* src, int dst, int n, float fill) {
    if (T.count >= CH_PACK_MAX) return;   // (83 entries today; the launcher checks the count)
    const int i = T.count++;
    T.src[i] = src;
    T.dst[i] = dst;
    T.n[i] = n;
    T.fill[i] = fill;
}

size_t vkn_chain_consts_floats() { return (size_t)2 * CA_TOTAL + CC_TOTAL; }

// out: [A block, raw-gather mode (dyn bias = bcnt scaled per row + dyn_b) | A block, x_feat mode (dyn bias = dyn_b) | C block]
int vkn_chain_pack_consts(const VknChainConsts& c, float* out, hipStream_t stream) {
    if (!out || c.ff <= 0 || c.ff > 2048 || c.ncls < 0 || c.ncls > 256) return VKN_E_ARG;
    ChPackTab T;
    T.count = 0;
    for (int mode = 0; mode < 2; ++mode) {
        const int o = mode * CA_TOTAL;
        ch_tab_add(T, mode == 0 ? c.bcnt : c.dyn_b, o + CA_DYN_B, 512, 0.f);
        ch_tab_add(T, mode == 0 ? c.dyn_b : nullptr, o + CA_DYN_B2, 512, 0.f);
        ch_tab_add(T, c.norm_out_w, o + CA_NO_W, 256, 1.f);   ch_tab_add(T, c.norm_out_b, o + CA_NO_B, 256, 0.f);
        ch_tab_add(T, c.inp_b, o + CA_INP_B, 512, 0.f);
        ch_tab_add(T, c.inorm_out_w, o + CA_INO_W, 256, 1.f); ch_tab_add(T, c.inorm_out_b, o + CA_INO_B, 256, 0.f);
        ch_tab_add(T, c.ig_b, o + CA_IG_B, 256, 0.f);
        ch_tab_add(T, c.inorm_in_w, o + CA_INI_W, 256, 1.f);  ch_tab_add(T, c.inorm_in_b, o + CA_INI_B, 256, 0.f);
        ch_tab_add(T, c.ug_b, o + CA_UG_B, 256, 0.f);
        ch_tab_add(T, c.norm_in_w, o + CA_NI_W, 256, 1.f);    ch_tab_add(T, c.norm_in_b, o + CA_NI_B, 256, 0.f);
        ch_tab_add(T, c.fc_b, o + CA_FC_B, 256, 0.f);
        ch_tab_add(T, c.fc_norm_w, o + CA_FCN_W, 256, 1.f);   ch_tab_add(T, c.fc_norm_b, o + CA_FCN_B, 256, 0.f);
        ch_tab_add(T, c.in_b, o + CA_IN_B, 768, 0.f);
        const float* sc[8] = {mode == 0 ? c.h2_inv[0] : c.h2_inv[1], c.h2_inv[2], c.h2_inv[3], c.h2_inv[4], c.h2_inv[5], c.h2_inv[6], nullptr, nullptr};
        for (int k = 0; k < 8; ++k) ch_tab_add(T, sc[k], o + CA_SC + k, 1, 1.f);
    }
    const int o = 2 * CA_TOTAL;
    ch_tab_add(T, c.out_b, o + CC_OUT_B, 256, 0.f);
    ch_tab_add(T, c.attn_norm_w, o + CC_AN_W, 256, 1.f);  ch_tab_add(T, c.attn_norm_b, o + CC_AN_B, 256, 0.f);
    ch_tab_add(T, c.ffn2_b, o + CC_B2, 256, 0.f);
    ch_tab_add(T, c.ffn_norm_w, o + CC_FN_W, 256, 1.f);   ch_tab_add(T, c.ffn_norm_b, o + CC_FN_B, 256, 0.f);
    ch_tab_add(T, c.cls_ln_w, o + CC_CLN_W, 256, 1.f);    ch_tab_add(T, c.cls_ln_b, o + CC_CLN_B, 256, 0.f);
    ch_tab_add(T, c.mask_ln_w, o + CC_MLN_W, 256, 1.f);   ch_tab_add(T, c.mask_ln_b, o + CC_MLN_B, 256, 0.f);
    ch_tab_add(T, c.dvec, o + CC_DVEC, 256, 0.f);
    ch_tab_add(T, c.fc_cls_b, o + CC_CLS_B, c.fc_cls_b ? c.ncls : 0, 0.f);
    ch_tab_add(T, nullptr, o + CC_CLS_B + (c.fc_cls_b ? c.ncls : 0), 256 - (c.fc_cls_b ? c.ncls : 0), 0.f);   // zero pad
    ch_tab_add(T, c.dec_b, o + CC_DEC_B, 256, 0.f);
    ch_tab_add(T, c.ffn1_b, o + CC_B1, c.ff, 0.f);
    ch_tab_add(T, nullptr, o + CC_B1 + c.ff, 2048 - c.ff, 0.f);
    for (int k = 0; k < 8; ++k) ch_tab_add(T, k < 7 ? c.h2_inv[7 + k] : nullptr, o + CC_SC + k, 1, 1.f);
    if (T.count >= CH_PACK_MAX) return VKN_E_ARG;
    hipLaunchKernelGGL(k_chain_pack, dim3(T.count), dim3(256), 0, stream, T, out);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

#else   // CH_H2: what only the fp16 form needs — its weight images
// fp32 W [Nout][K] -> fp16 hi / lo tile images Wp[ceil(Nout/256)][K/32][2][4][256][8] of W * *scale (rows >= Nout zero), the layout of
// k_split_w3 (vkn_update.hip) with two planes.  scale: DEVICE scalar, a power of two (vkn_pow2_scale_f32 of the matrix: its maximum at
// 2^9 .. 2^10, so that the low half of every weight down to 2^-12 of the maximum keeps its eleven bits); grid = (K/32, ceil(Nout/256)).
__global__ __launch_bounds__(256) void k_split_h2(const float* __restrict__ W, _Float16* __restrict__ Wp, int Nout, int K,
                                                  const float* __restrict__ scale) {
    const int kt = blockIdx.x, nt = blockIdx.y;
    _Float16* dst = Wp + ((size_t)nt * gridDim.x + kt) * (CH_WTILE / 2);
    const int row = threadIdx.x, n = nt * 256 + row;
    const float sc = scale[0];
    for (int k = 0; k < 32; ++k) {
        ch_e t[2] = {(ch_e)0.f, (ch_e)0.f};
        if (n < Nout) ch_split(W[(size_t)n * K + kt * 32 + k] * sc, t);
        const int q = k >> 3, e = k & 7;
        dst[((0 * 4 + q) * 256 + row) * 8 + e] = t[0];
        dst[((1 * 4 + q) * 256 + row) * 8 + e] = t[1];
    }
}
size_t vkn_split_h2_bytes(int Nout, int K) { return (size_t)((Nout + 255) / 256) * (size_t)(K / 32) * CH_WTILE; }
int vkn_launch_split_h2(const float* W, void* images, int Nout, int K, const float* scale, hipStream_t stream) {
    if (!W || !images || !scale || Nout <= 0 || K <= 0 || K % 32 != 0) return VKN_E_ARG;
    hipLaunchKernelGGL(k_split_h2, dim3(K / 32, (Nout + 255) / 256), dim3(256), 0, stream, W, static_cast<_Float16*>(images), Nout, K, scale);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
#endif  // CH_H2

int vkn_launch_chain_a(const VknChainA& p, hipStream_t stream) {
    if (!p.a0 || !p.obj_in || !p.wbase || !p.obj1 || !p.qkv || !p.consts || p.M <= 0 || p.wbytes >= (1ull << 31)) return VKN_E_ARG;
    ChainAArgs A{};
    A.a0 = p.a0; A.obj_in = p.obj_in; A.rowscale = p.rowscale; A.wbase = p.wbase; A.wbytes = (unsigned)p.wbytes;
    A.off_dyn = p.off_dyn; A.off_inp = p.off_inp; A.off_ig = p.off_ig; A.off_ug = p.off_ug; A.off_fc = p.off_fc; A.off_in = p.off_in;
    A.eps = p.eps; A.M = p.M; A.obj1 = p.obj1; A.qkv = p.qkv;
    A.consts = p.consts + (p.rowscale ? 0 : CA_TOTAL);
    A.status = p.status;
#ifdef CH_H2
    const size_t lds = (size_t)2 * CH_IMG * sizeof(ch_e) + (size_t)(2 * CH_SBUF + CA_TOTAL + 16 * CH_THREADS + 64 + 512) * sizeof(float);
#else
    const size_t lds = (size_t)2 * CH_IMG * sizeof(ch_e) + (size_t)(2 * CH_SBUF + CA_TOTAL + 16 * CH_THREADS) * sizeof(float);
#endif
#define CHA_LAUNCH(ABLV)                                                                                            \
    do {                                                                                                            \
        VKN_ALLOW_FULL_LDS(k_chain_a<ABLV>);                                                                        \
        hipLaunchKernelGGL(k_chain_a<ABLV>, dim3((p.M + CH_ROWS - 1) / CH_ROWS), dim3(CH_THREADS), lds, stream, A); \
    } while (0)
#ifdef VKN_DEBUG
    switch (vkn_dbg_env("VKN_CHAIN_ABL", 0)) {
        case 1: CHA_LAUNCH(1); break;
        case 2: CHA_LAUNCH(2); break;
        case 3: CHA_LAUNCH(3); break;
        case 4: CHA_LAUNCH(4); break;
        case 5: CHA_LAUNCH(5); break;
        case 6: CHA_LAUNCH(6); break;
        case 7: CHA_LAUNCH(7); break;
        case 8: CHA_LAUNCH(8); break;
        case 9: CHA_LAUNCH(9); break;
        case 10: CHA_LAUNCH(10); break;
        case 11: CHA_LAUNCH(11); break;
        default: CHA_LAUNCH(0); break;
    }
#else
    CHA_LAUNCH(0);
#endif
#undef CHA_LAUNCH
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

int vkn_launch_chain_c(const VknChainC& p, hipStream_t stream) {
    if (!p.ao || !p.obj1 || !p.wbase || !p.obj_out || !p.consts || p.M <= 0 || p.wbytes >= (1ull << 31)) return VKN_E_ARG;
    if (p.ff <= 0 || p.ff % 256 != 0 || p.ff > 2048 || p.ncls > 256) return VKN_E_SHAPE;
    if (!p.plane_hi == !p.kern_out || (p.plane_hi && !p.plane_lo)) return VKN_E_ARG;   // exactly one output form of the decode kernels
    ChainCArgs A{};
    A.ao = p.ao; A.obj1 = p.obj1; A.wbase = p.wbase; A.wbytes = (unsigned)p.wbytes;
    A.off_out = p.off_out; A.off_ffn1 = p.off_ffn1; A.off_ffn2 = p.off_ffn2; A.off_clsfc = p.off_clsfc; A.off_maskfc = p.off_maskfc;
    A.off_fccls = p.off_fccls; A.off_dec = p.off_dec;
    A.nchunks = p.ff / 256; A.has_cls = (p.cls_out != nullptr) ? 1 : 0; A.cls_sigmoid = p.cls_sigmoid; A.ncls = p.ncls;
    A.eps = p.eps; A.M = p.M; A.kb0 = p.kb0; A.obj_out = p.obj_out; A.cls_out = p.cls_out; A.kb_out = p.kb_out;
    A.plane_hi = p.plane_hi; A.plane_lo = p.plane_lo; A.kern_out = p.kern_out; A.rows_per_frame = p.rows_per_frame; A.NPT = p.NPT;
    A.consts = p.consts + 2 * CA_TOTAL;
    A.status = p.status;
#ifdef CH_H2
    const size_t lds = (size_t)2 * CH_IMG * sizeof(ch_e) + (size_t)(2 * CH_SBUF + CC_TOTAL + 32) * sizeof(float);
#else
    const size_t lds = (size_t)2 * CH_IMG * sizeof(ch_e) + (size_t)(2 * CH_SBUF + CC_TOTAL) * sizeof(float);
#endif
#define CHC_LAUNCH(ABLV)                                                                                            \
    do {                                                                                                            \
        VKN_ALLOW_FULL_LDS(k_chain_c<ABLV>);                                                                        \
        hipLaunchKernelGGL(k_chain_c<ABLV>, dim3((p.M + CH_ROWS - 1) / CH_ROWS), dim3(CH_THREADS), lds, stream, A); \
    } while (0)
#ifdef VKN_DEBUG
    switch (vkn_dbg_env("VKN_CHAIN_ABL", 0)) {
        case 1: CHC_LAUNCH(1); break;
        case 2: CHC_LAUNCH(2); break;
        case 3: CHC_LAUNCH(3); break;
        case 4: CHC_LAUNCH(4); break;
        case 5: CHC_LAUNCH(5); break;
        case 6: CHC_LAUNCH(6); break;
        case 7: CHC_LAUNCH(7); break;
        case 8: CHC_LAUNCH(8); break;
        case 9: CHC_LAUNCH(9); break;
        case 10: CHC_LAUNCH(10); break;
        case 11: CHC_LAUNCH(11); break;
        default: CHC_LAUNCH(0); break;
    }
#else
    CHC_LAUNCH(0);
#endif
#undef CHC_LAUNCH
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
