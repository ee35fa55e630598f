// vkn_ksplit.hip — the [B*N, C] "kernel update + interaction" chain for FEW rows (one to ~sixteen frames per call: the regime a clip
// of 8 frames sharded over 8 GPUs runs, and the reference's own one-frame-per-call inference loop).  One launch per GEMM PHASE, but
// each GEMM is spread over the chip by COLUMN blocks and its contraction over the eight waves of a workgroup, so that a launch is one
// memory round trip + a dozen MFMAs per wave instead of a 4 us weight stream through four workgroups (k_gemm_t3, vkn_chain.hip).
//
// Covers (reference file:line) — the same functions as vkn_chain.hip:
//   KernelUpdator.forward                      knet/kernel_updator.py:56-93
//   attention_norm(attention(obj)) in/out_proj knet/det/kernel_update_head.py:204-208
//   ffn_norm(ffn(obj))                         knet/det/kernel_update_head.py:214-215
//   cls_fcs / fc_cls / mask_fcs / fc_mask      knet/det/kernel_update_head.py:217-227
//   link block LN(FFN(LN(cur + MHA(cur, kv)))) knet/video/kernel_update_head.py:394-415
//
// Design (DESIGN.md §5 "few-row chain"):
//   * Workgroup = 32 MT rows x 32 NCB output columns, 8 waves; wave w owns K-tile(s) w of the 256 KPW-wide contraction chunk: it
//     requests ITS six weight fragments per (column block, K-tile) — the pre-split bf16x3 tile images of vkn_prepare_stage_f32,
//     already the MFMA operand layout — and ITS 32-wide slice of the activation rows straight from global memory into registers, all
//     at once: ONE round trip, no LDS staging, no K loop.  117 rows x 256 columns = 32 workgroups instead of 4; the FFN 256.
//   * Row-wise normalisation moves from the PRODUCER's epilogue (which would need whole rows in one workgroup) to the CONSUMER's
//     prologue: a GEMM stores its raw output (+ bias, + residual) and the next GEMM applies LayerNorm / ReLU / sigmoid / the updator's
//     gate and mix algebra to the rows it reads anyway.  Row statistics: 16 partials per row (8 waves x 2 half-waves) through 2 KB of
//     LDS, summed in fixed order by every lane — the exchange of vkn_chain.hip.  Normalised rows that are ALSO results (updated
//     kernels, the stage's output kernels, residuals) are written once, by the workgroups of column group 0 ("side output").
//   * The eight partial accumulators of a tile meet in LDS (fixed order, deterministic); the epilogue (bias, count-scaled bias,
//     residual, activation, fp32 / f16-plane stores) runs on two adjacent columns per thread.
//   * A contraction longer than 256 (the FFN's second Linear, K = ff) is split over blockIdx.z in chunks of 256 KPW; the partial
//     results are summed (fixed order) by the consumer's prologue (NS summands), never by atomics.
//   * Arithmetic: the six significant cross products of the bf16x3 split per operand pair, smallest first, fp32 accumulation — as
//     k_gemm_s3 / k_gemm_t3 / k_chain_* (2^-24 relative).  Summation ORDER differs from those forms (K-tiles meet in LDS; LayerNorm
//     partials run over k-slices): results agree to fp32 rounding, not bit for bit.  LayerNorm two-pass in fp32.
//
// Why launches and not one cooperative kernel with grid barriers: MI355X_MICROARCH.md prices a dependent kernel boundary at 1.45 us
// and the cheapest device-wide barrier (barrier-xcd) at 4.1-4.8 us + an L2 write-back; a phase here IS one round trip, so the
// boundary is the cheaper seam.
#include "vkn_common.h"
#include "vkn_launch.h"

// Floating-point contraction is OFF in this file: the same phase runs in several tile shapes (NCB, MT) picked by row count, and hipcc
// contracts `a * b + c` into an fma in one instantiation and not in another (seen: the count-scaled bias `bias * count + bias2` of the
// dynamic layer, the decode-bias dot product) — a clip cut into blocks would then differ from the whole clip in the last bit.  Every
// prologue / epilogue value is the same sequence of individually rounded fp32 operations in every instantiation; the MFMAs are unaffected.
#pragma clang fp contract(off)

typedef __bf16 kbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ku32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 khalf2 __attribute__((ext_vector_type(2)));

#define KS_THREADS 512
#define KS_WTILE 49152u
#define KS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// pin a loaded value HERE: hipcc otherwise sinks a load whose only use sits behind a (uniform) branch into that branch — behind the
// wait for the activation rows, i.e. a second memory round trip for every optional vector of the prologue / epilogue
#define KS_PIN(x) asm volatile("" : "+v"(x))

namespace {

__device__ __forceinline__ void ks_split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// this lane's 16 values of a 32-wide k-slice: p points at (row, slice base + 8 g); k-step 0 = p[0..7], k-step 1 = p[16..23]
__device__ __forceinline__ void ks_load16(const float* __restrict__ p, float (&o)[16]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p + 16), d = *reinterpret_cast<const f32x4*>(p + 20);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = a[e];
        o[4 + e] = b[e];
        o[8 + e] = c[e];
        o[12 + e] = d[e];
    }
}
__device__ __forceinline__ void ks_store16(float* __restrict__ p, const float (&v)[16]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p + 16) = f32x4{v[8], v[9], v[10], v[11]};
    *reinterpret_cast<f32x4*>(p + 20) = f32x4{v[12], v[13], v[14], v[15]};
}

// sum over the 256 columns of a row for NV per-lane partials: 16 partials per row (8 waves x 2 half-waves) through LDS, summed in
// fixed order by every lane of the row.  One barrier.
template <int NV>
__device__ __forceinline__ void ks_rowsum(float (&p)[NV], float* Sbuf, int wave, int g, int li) {
#pragma unroll
    for (int v = 0; v < NV; ++v) Sbuf[v * 512 + (wave * 2 + g) * 32 + li] = p[v];
    KS_BAR();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += Sbuf[v * 512 + k * 32 + li];
        p[v] = s;
    }
}

// LayerNorm statistics of NV row vectors held as 16 values per lane (16 such slices per row): every slice computes ITS mean and ITS
// centred sum of squares (two-pass, in registers), ONE exchange of the 16 (mean, M2) pairs per row, then the pairwise-merge identity
//   mean = sum_i mean_i / 16,   M2 = sum_i M2_i + 16 sum_i (mean_i - mean)^2        (equal slice sizes)
// — the parallel form of the two-pass variance (what torch's own LayerNorm kernels merge per thread), not E[x^2] - mean^2: no
// cancellation.  Fixed summation order; one barrier (the two-exchange form cost ~0.7 us more per phase).
template <int NV>
__device__ __forceinline__ void ks_ln_stats(const float (&v)[NV][16], float eps, float* S, int wave, int g, int li, float (&mean)[NV],
                                            float (&rstd)[NV]) {
    float p[2 * NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[i][r];
        const float m = s * (1.0f / 16.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = v[i][r] - m;
            q += d * d;
        }
        p[2 * i] = m;
        p[2 * i + 1] = q;
    }
#pragma unroll
    for (int k = 0; k < 2 * NV; ++k) S[k * 512 + (wave * 2 + g) * 32 + li] = p[k];
    KS_BAR();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float mi[16], ms = 0.f, qs = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            mi[k] = S[(2 * i) * 512 + k * 32 + li];
            ms += mi[k];
            qs += S[(2 * i + 1) * 512 + k * 32 + li];
        }
        mean[i] = ms * (1.0f / 16.f);
        float dm = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float d = mi[k] - mean[i];
            dm += d * d;
        }
        rstd[i] = 1.0f / sqrtf((qs + 16.f * dm) * (1.0f / 256.f) + eps);
    }
}

// 24 zero floats: the stand-in operand of an absent optional vector (its loads stay unconditional: no branch, no extra scalar-load
// round trip per optional operand in the kernel prologue)
__device__ __attribute__((aligned(16))) float ks_zeros[32] = {0.f};   // (never written; not `const`, or hipcc folds the loads back into branches)

__device__ __forceinline__ float ks_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// the 32 gate sigmoids per lane of the updator's mix: v_exp_f32 + v_rcp_f32 (1 ulp each; |error| < 2e-7 on a value in (0, 1)) instead of
// the ~22-instruction expf + IEEE division pair — the mix phase is instruction-issue bound (700 of its 2000 VALU instructions)
__device__ __forceinline__ float ks_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

}  // namespace

// MODE 0: A = act(LN(sum_s a0[s] + pbias + presid))      (every piece optional; NS = 1 or up to 4 summands)
// MODE 1: A = a0 .* a1                                    (the updator's gate: input_in * parameters_in)
// MODE 2: A = sigmoid(LN1(a1)) .* LN2(a2) + sigmoid(LN0(a0)) .* LN3(a3)      (the updator's mix: update_gate * param_out + input_gate * input_out)
// ZS: blockIdx.z selects a chunk of the contraction (one problem) instead of the problem — a template parameter so that the problem
// descriptor's address depends on nothing but blockIdx (its scalar loads start with the kernel, not behind a first kernarg load)
template <int MODE, int NCB, int MT, int KPW, int NS, bool ZS>
__global__ __launch_bounds__(KS_THREADS) void k_gemm_ks(const VknKsProb p0, const VknKsProb p1, int M, long long out_zstride) {
    constexpr int NACC = NCB * MT;
    extern __shared__ __attribute__((aligned(16))) char smem_ks[];
    float* RED = reinterpret_cast<float*>(smem_ks);   // [8 waves][NACC][16][64]
    float* S = RED + 8 * NACC * 1024;                 // [8][512] LayerNorm statistics exchange + [2][512] dot-product exchange
    float* LNC = S + 10 * 512;                        // MODE 2: [8][256] LayerNorm vectors

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const bool second = (!ZS) && (blockIdx.z == 1);
    const VknKsProb& P = second ? p1 : p0;
    const VknKsPro& R = P.pro;
    const VknKsEpi& E = P.epi;
    const int kz = ZS ? (int)blockIdx.z : 0;
    const int cb0 = blockIdx.x * NCB;
    if (cb0 * 32 >= P.Nout) return;   // grouped launch: the grid is sized for the wider problem (uniform exit before any barrier)
    const int m0 = blockIdx.y * (32 * MT);

    // ---- (1) this wave's weight fragments: column blocks cb0 .. cb0 + NCB - 1, K-tiles (kz 8 + wave) KPW + j
    ku32x4 wf[NCB][KPW][6];
    {
        const int ntile = (P.Nout + 255) >> 8;
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.Wsplit), 0, (int)((unsigned)(ntile * P.KT) * KS_WTILE), 0x00020000);
        const unsigned voff = (unsigned)(g * 4096 + ((cb0 & 7) * 32 + li) * 16);
        const unsigned tbase = (unsigned)((cb0 >> 3) * P.KT + (kz * 8 + wave) * KPW) * KS_WTILE;
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int j = 0; j < KPW; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        wf[c][j][ks * 3 + p] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(voff + (unsigned)c * 512u),
                                                                                     (int)(tbase + (unsigned)j * KS_WTILE + (unsigned)((p * 4 + 2 * ks) * 4096)), 0);
    }

    int rowc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rowc[mt] = min(m0 + 32 * mt + li, M - 1);
    // ---- (1b) epilogue operands of this thread's two adjacent columns per accumulator (requested with the weights: their round trip
    //      rides under the prologue)
    //      thread (wave w', lane) finishes registers 2 w', 2 w' + 1: row li, columns 8 (w' >> 1) + 4 g + 2 (w' & 1) + {0, 1} of a block
    const int ecol = 8 * (wave >> 1) + 4 * g + 2 * (wave & 1);
    float eb[NCB][2], er[NACC][2], ebs[MT];
    {
        // absent operands read ks_zeros[0] (index multiplier 0): unconditional loads, no branch per optional operand
        const bool hb = E.bias != nullptr, hb2 = E.bias2 != nullptr, hr = E.resid != nullptr, hs = hb && E.rowscale;
        const float *bp = hb ? E.bias : ks_zeros, *b2p = hb2 ? E.bias2 : ks_zeros, *rp = hr ? E.resid : ks_zeros, *sp = hs ? E.rowscale : ks_zeros;
        const int mb = hb ? 1 : 0, mb2 = hb2 ? 1 : 0, mr = hr ? 1 : 0, ms = hs ? 1 : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float rsv = sp[rowc[mt] * ms];
            ebs[mt] = hs ? rsv : 1.f;
        }
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int col = min((cb0 + c) * 32 + ecol + e, P.Nout - 1);
                eb[c][e] = bp[col * mb];
                const float b2v = b2p[col * mb2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) er[mt * NCB + c][e] = rp[((size_t)rowc[mt] * E.ldr + col) * mr] + b2v;
            }
    }

    // ---- (2) the activation slice of this wave: rows m0 + 32 mt + li, columns kz 256 KPW + wave 32 KPW + 32 j + 16 ks + 8 g + e
    const int kcol = kz * 256 * KPW + wave * 32 * KPW + 8 * g;   // first column of this lane's slice (j = 0, ks = 0)
    float av[MT][KPW][16];
    const bool sideg = (blockIdx.x == 0);   // the column group that also writes the prologue's side outputs

    if (MODE == 0) {
        float t[NS][MT][KPW][16];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s == 0 || s < R.nsum) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < KPW; ++j)
                        ks_load16(R.a[0] + (size_t)s * R.sum_stride + (size_t)rowc[mt] * R.lda[0] + kcol + 32 * j, t[s][mt][j]);
            }
        float pb[16], pr[MT][16], lw[16], lb[16], dv[16];
        const bool has_pb = (KPW == 1) && R.pbias, has_pr = (KPW == 1) && R.presid, has_ln = (KPW == 1) && R.ln_w[0];
        const bool has_dot = (KPW == 1) && R.dot_out && sideg;
        if (KPW == 1) {   // optional vectors: the loads are unconditional (absent -> 24 zero floats), only their USE is conditional
            ks_load16(has_pb ? R.pbias + kcol : ks_zeros, pb);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ks_load16(has_pr ? R.presid + (size_t)rowc[mt] * R.ldr + kcol : ks_zeros, pr[mt]);
            ks_load16(has_ln ? R.ln_w[0] + kcol : ks_zeros, lw);
            ks_load16(has_ln ? R.ln_b[0] + kcol : ks_zeros, lb);
            ks_load16(has_dot ? R.dot_vec + kcol : ks_zeros, dv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                KS_PIN(pb[r]);
                KS_PIN(lw[r]);
                KS_PIN(lb[r]);
                KS_PIN(dv[r]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) KS_PIN(pr[mt][r]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) av[mt][j][r] = t[0][mt][j][r];
#pragma unroll
                for (int s = 1; s < NS; ++s)
                    if (s < R.nsum) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) av[mt][j][r] += t[s][mt][j][r];
                    }
            }
        if (KPW == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) av[mt][0][r] += pb[r] + pr[mt][r];   // (zeros when absent)
            if (has_ln) {
                float vv[MT][16], mean[MT], rstd[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) vv[mt][r] = av[mt][0][r];
                ks_ln_stats<MT>(vv, R.eps, S, wave, g, li, mean, rstd);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) av[mt][0][r] = (vv[mt][r] - mean[mt]) * rstd[mt] * lw[r] + lb[r];
            }
            if (R.act == 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) av[mt][0][r] = fmaxf(av[mt][0][r], 0.f);
            }
            if (R.side_out && sideg) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (m0 + 32 * mt + li < M) ks_store16(R.side_out + (size_t)(m0 + 32 * mt + li) * R.ld_side + kcol, av[mt][0]);
            }
            if (has_dot) {   // side output dot_out[row] = A[row] . dot_vec (+ *dot_bias): the folded decode bias
                float pd[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += av[mt][0][r] * dv[r];
                    pd[mt] = s;
                }
                ks_rowsum<MT>(pd, S + 8 * 512, wave, g, li);   // (its own exchange buffer: a slower wave may still read the LayerNorm's)
                if (wave == 0 && g == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (m0 + 32 * mt + li < M) R.dot_out[m0 + 32 * mt + li] = pd[mt] + (R.dot_bias ? *R.dot_bias : 0.f);
                }
            }
        }
    } else if (MODE == 1) {
        float x[16], y[16];
        ks_load16(R.a[0] + (size_t)rowc[0] * R.lda[0] + kcol, x);
        ks_load16(R.a[1] + (size_t)rowc[0] * R.lda[1] + kcol, y);
#pragma unroll
        for (int r = 0; r < 16; ++r) av[0][0][r] = x[r] * y[r];
    } else {
        // the eight LayerNorm vectors -> LDS (one 16-byte load per thread), the four operands -> registers
        {
            const int vec = tid >> 6, q = tid & 63;   // vector 0..7 = (w0, b0, w1, b1, ..), 64 float4 each
            const float* src = (vec & 1) ? R.ln_b[vec >> 1] : R.ln_w[vec >> 1];
            *reinterpret_cast<f32x4*>(LNC + vec * 256 + 4 * q) = *reinterpret_cast<const f32x4*>(src + 4 * q);
        }
        float vv[4][16], mean[4], rstd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ks_load16(R.a[i] + (size_t)rowc[0] * R.lda[i] + kcol, vv[i]);
        ks_ln_stats<4>(vv, R.eps, S, wave, g, li, mean, rstd);   // (its barriers publish LNC)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lw[16], lb[16];
            ks_load16(LNC + (2 * i) * 256 + wave * 32 + 8 * g, lw);
            ks_load16(LNC + (2 * i + 1) * 256 + wave * 32 + 8 * g, lb);
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[i][r] = (vv[i][r] - mean[i]) * rstd[i] * lw[r] + lb[r];
            // one vector pair at a time (hoisting all eight LDS reads costs 128 VGPRs -> spills): the results are pinned here, the
            // next pair's reads stay behind the fence
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(vv[i][r]));
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) av[0][0][r] = ks_sigmoid_fast(vv[1][r]) * vv[2][r] + ks_sigmoid_fast(vv[0][r]) * vv[3][r];
    }

#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            KS_PIN(eb[c][e]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) KS_PIN(er[mt * NCB + c][e]);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) KS_PIN(ebs[mt]);

    // ---- (4) split the activation slice, 12 MFMAs per (row block, column block, K-tile)
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kbf16x8 ah, am, al;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 hh, mm, ll;
                    ks_split3(av[mt][j][8 * ks + e], hh, mm, ll);
                    ah[e] = hh;
                    am[e] = mm;
                    al[e] = ll;
                }
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const kbf16x8 wh = __builtin_bit_cast(kbf16x8, wf[c][j][ks * 3 + 0]);
                    const kbf16x8 wm = __builtin_bit_cast(kbf16x8, wf[c][j][ks * 3 + 1]);
                    const kbf16x8 wl = __builtin_bit_cast(kbf16x8, wf[c][j][ks * 3 + 2]);
                    f32x16& A = acc[mt * NCB + c];
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, A, 0, 0, 0);
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, A, 0, 0, 0);
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, am, A, 0, 0, 0);
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, ah, A, 0, 0, 0);
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, am, A, 0, 0, 0);
                    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, A, 0, 0, 0);
                }
            }

    // ---- (5) the eight K-slices of a tile meet in LDS, fixed order
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) RED[((wave * NACC + a) * 16 + r) * 64 + lane] = acc[a][r];
    KS_BAR();
    float* outz = E.out ? E.out + (size_t)kz * out_zstride : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + 32 * mt + li;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int a = mt * NCB + c;
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float s = 0.f;
#pragma unroll
                for (int src = 0; src < 8; ++src) s += RED[((src * NACC + a) * 16 + 2 * wave + e) * 64 + lane];
                s += eb[c][e] * ebs[mt] + er[a][e];
                if (E.act == 1) s = fmaxf(s, 0.f);
                else if (E.act == 2) s = ks_sigmoid(s);
                v[e] = s;
            }
            if (row >= M) continue;
            const int col = (cb0 + c) * 32 + ecol;
            if (outz) {
                float* o = outz + (size_t)row * E.ldo + col;
                if (col + 1 < P.Nout && (E.ldo & 1) == 0 && (reinterpret_cast<uintptr_t>(outz) & 7) == 0) {
                    *reinterpret_cast<f32x2*>(o) = f32x2{v[0], v[1]};
                } else {
                    if (col < P.Nout) o[0] = v[0];
                    if (col + 1 < P.Nout) o[1] = v[1];
                }
            }
            if (E.plane_hi) {   // f16 split planes [B][NPT][ldo]: row r = b N + n -> plane row b NPT + n  (Nout % 2 == 0 here)
                const int fb = row / E.rows_per_frame, n = row - fb * E.rows_per_frame;
                const size_t base = ((size_t)fb * E.NPT + n) * E.ldo + col;
                if (col + 1 < P.Nout) {
                    khalf2 h, l;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        _Float16 hh, ll;
                        vkn_split_f16(v[e], hh, ll);
                        h[e] = hh;
                        l[e] = ll;
                    }
                    *reinterpret_cast<khalf2*>(E.plane_hi + base) = h;
                    *reinterpret_cast<khalf2*>(E.plane_lo + base) = l;
                }
            }
        }
    }
}

// One GEMM phase of the few-row chain.  nprob 1 or 2 problems (same M, K = 256 each; blockIdx.z selects) — or, zchunks > 0, ONE problem
// whose contraction of zchunks x 256 kpw is split over blockIdx.z (partial results at out + z out_zstride; no prologue extras).
// VKN_E_SHAPE = not applicable.
int vkn_launch_gemm_ks(const VknKsProb* probs, int nprob, int mode, int zchunks, int kpw, long long out_zstride, int M, hipStream_t stream) {
    if (!probs || nprob < 1 || nprob > 2 || M <= 0 || mode < 0 || mode > 2) return VKN_E_ARG;
    if (zchunks < 0 || (zchunks > 0 && (nprob != 1 || mode != 0)) || (kpw != 1 && kpw != 2) || (kpw == 2 && zchunks == 0)) return VKN_E_ARG;
    int nmax = 0, ns = 1;
    for (int i = 0; i < nprob; ++i) {
        const VknKsProb& p = probs[i];
        if (!p.Wsplit || p.Nout <= 0 || p.KT < 8 * kpw * (zchunks > 0 ? zchunks : 1)) return VKN_E_ARG;
        if ((size_t)((p.Nout + 255) / 256) * p.KT * KS_WTILE >= (1ull << 31)) return VKN_E_SHAPE;
        const int nop = mode == 0 ? 1 : (mode == 1 ? 2 : 4);
        for (int k = 0; k < nop; ++k)
            if (!p.pro.a[k] || (p.pro.lda[k] & 3) || (reinterpret_cast<uintptr_t>(p.pro.a[k]) & 15)) return VKN_E_SHAPE;
        if (mode == 0 && (p.pro.nsum < 1 || p.pro.nsum > 4 || (p.pro.nsum > 1 && (p.pro.sum_stride & 3)))) return VKN_E_ARG;
        if (mode == 2)
            for (int k = 0; k < 4; ++k)
                if (!p.pro.ln_w[k] || !p.pro.ln_b[k] || (reinterpret_cast<uintptr_t>(p.pro.ln_w[k]) & 15) || (reinterpret_cast<uintptr_t>(p.pro.ln_b[k]) & 15)) return VKN_E_SHAPE;
        if (mode == 0) {
            const VknKsPro& r = p.pro;
            if (zchunks > 0 && (r.nsum != 1 || r.pbias || r.presid || r.ln_w[0] || r.side_out || r.dot_out || r.act)) return VKN_E_ARG;
            if ((r.ln_w[0] && (!r.ln_b[0] || ((reinterpret_cast<uintptr_t>(r.ln_w[0]) | reinterpret_cast<uintptr_t>(r.ln_b[0])) & 15))) ||
                (r.pbias && (reinterpret_cast<uintptr_t>(r.pbias) & 15)) || (r.presid && ((reinterpret_cast<uintptr_t>(r.presid) & 15) || (r.ldr & 3))) ||
                (r.side_out && ((reinterpret_cast<uintptr_t>(r.side_out) & 15) || (r.ld_side & 3))) ||
                (r.dot_out && (!r.dot_vec || (reinterpret_cast<uintptr_t>(r.dot_vec) & 15))))
                return VKN_E_SHAPE;
            if (r.nsum > ns) ns = r.nsum;
        }
        if (p.epi.plane_hi && (!p.epi.plane_lo || (p.Nout & 1) || (p.epi.ldo & 1) || p.epi.rows_per_frame <= 0)) return VKN_E_ARG;
        nmax = p.Nout > nmax ? p.Nout : nmax;
    }
    const int rt = (M + 31) / 32, cbs = (nmax + 31) / 32, nz = zchunks > 0 ? zchunks : nprob;
    // tile shape: at most ONE workgroup per CU (256) — a second round of workgroups costs a whole extra round trip, a fatter tile only
    // more MFMAs behind the same one (chain alone, us per stage at 234 / 351 / 468 / 585 / 936 rows: cap 768: 83 / 101 / 99 / 140 / 142,
    // cap 256: 78 / 87 / 87 / 109 / 117, cap 128: 83 / 100 / 99 / 117 / 122; profiles/r05_chain_forms.txt)
    int ncb = 1, mt = 1;
    if (mode == 0) {
        auto wgs = [&](int c, int m) { return (long long)((cbs + c - 1) / c) * ((rt + m - 1) / m) * nz; };
        const long long cap = vkn_dbg_env("VKN_KS_WGCAP", 256);   // (debug build: A/B of the workgroup-count cap)
        if (wgs(1, 1) > cap && cbs >= 2) ncb = 2;
        if (ncb == 2 && wgs(2, 1) > cap && ns == 1) mt = 2;
    }
    const dim3 grid((cbs + ncb - 1) / ncb, (rt + mt - 1) / mt, nz);
    const size_t lds = (size_t)(8 * ncb * mt * 1024 + 10 * 512 + (mode == 2 ? 8 * 256 : 0)) * sizeof(float);
    const VknKsProb& q1 = probs[nprob > 1 ? 1 : 0];
#define KS_LAUNCH_(MODEV, NCBV, MTV, KPWV, NSV, ZSV)                                                                                \
    do {                                                                                                                            \
        if (lds > 64 * 1024) VKN_ALLOW_FULL_LDS((k_gemm_ks<MODEV, NCBV, MTV, KPWV, NSV, ZSV>));                                     \
        hipLaunchKernelGGL((k_gemm_ks<MODEV, NCBV, MTV, KPWV, NSV, ZSV>), grid, dim3(KS_THREADS), lds, stream, probs[0], q1, M,     \
                           out_zstride);                                                                                            \
    } while (0)
    const bool zs = zchunks > 0;
    if (mode == 1) KS_LAUNCH_(1, 1, 1, 1, 1, false);
    else if (mode == 2) KS_LAUNCH_(2, 1, 1, 1, 1, false);
    else if (ns > 1) {
        if (ncb == 1) KS_LAUNCH_(0, 1, 1, 1, 4, false);
        else KS_LAUNCH_(0, 2, 1, 1, 4, false);
    } else if (kpw == 2) {
        if (ncb == 1) KS_LAUNCH_(0, 1, 1, 2, 1, true);
        else if (mt == 1) KS_LAUNCH_(0, 2, 1, 2, 1, true);
        else KS_LAUNCH_(0, 2, 2, 2, 1, true);
    } else if (zs) {
        if (ncb == 1) KS_LAUNCH_(0, 1, 1, 1, 1, true);
        else if (mt == 1) KS_LAUNCH_(0, 2, 1, 1, 1, true);
        else KS_LAUNCH_(0, 2, 2, 1, 1, true);
    } else {
        if (ncb == 1) KS_LAUNCH_(0, 1, 1, 1, 1, false);
        else if (mt == 1) KS_LAUNCH_(0, 2, 1, 1, 1, false);
        else KS_LAUNCH_(0, 2, 2, 1, 1, false);
    }
#undef KS_LAUNCH_
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
