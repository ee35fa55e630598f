// vkn_common.h — shared device helpers for the MI355X (gfx950 / CDNA4) kernel-update-head kernels.
// wave = 64 lanes; MFMA 32x32 fragments; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define VKN_WAVE 64

// Debug knobs: the release library reads NO environment variable and contains no ablation variant; a build with -DVKN_DEBUG
// (tools/ only, `_lib.build_debug()` -> lib/libvkn_debug.so) turns `vkn_dbg_env(name, default)` into a getenv read.
#ifdef VKN_DEBUG
#include <stdlib.h>
static inline int vkn_dbg_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
#define vkn_dbg_env(name, dflt) (dflt)
#endif
// Time-attribution / ablation arms of the shipped kernels (template parameter ABL, upsample NT 2 / 3): `VKN_ABL_IS(ABL, k)` is the
// comparison in the debug build and the literal `false` in the release build — the preprocessor removes the arms from the
// release kernels instead of leaving dead template branches in the product.
#ifdef VKN_DEBUG
#define VKN_ABL_IS(param, k) ((param) == (k))
#else
#define VKN_ABL_IS(param, k) false
#endif

// ---- MFMA 32x32 C/D fragment map (dtype independent on gfx950): lane l, register r ->
//      col = l & 31 ; row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ int vkn_cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// f16 two-term split: v ~= hi + lo with |v - hi - lo| <= 2^-22 |v| (RNE both), products hi*hi, hi*lo, lo*hi
// are exact in the fp32 MFMA accumulator.  Valid for |v| < 65504.
__device__ __forceinline__ void vkn_split_f16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// The same split for TWO values in four VALU operations instead of eight: v_cvt_pk_f16_f32 (RNE, both highs in one dword),
// two v_fma_mix_f32 (r = v - float(hi): the f16 half is an operand, no separate conversion; exact like the subtraction),
// v_cvt_pk_f16_f32 again.  Bit-identical to two vkn_split_f16 calls.
typedef _Float16 vkn_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void vkn_split_f16x2(float a, float b, vkn_half2& hi, vkn_half2& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_convertvector(v, vkn_half2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(b));
    const f32x2 r = {r0, r1};
    lo = __builtin_convertvector(r, vkn_half2);
}

// Wave-wide sum / max, the same value in every lane.  DPP row shifts inside the 16-lane rows, then the two row broadcasts (gfx9):
// six dependent VALU operations (~60 cycles) instead of six ds_bpermute round trips (~700 cycles) — the row epilogue of every
// [N x C] GEMM runs eight of these reductions back to back (LayerNorm of four rows per wave), 2.4 of its 3.4 us before this.
// Summation order: inclusive scan by 1, 2, 4, 8 inside each row of 16, rows 0+1 and 2+3, then the halves — fixed, deterministic.
#define VKN_DPP_STEP(OP, X, CTRL, ROWMASK, IDENT)                                                                            \
    X = OP(X, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(IDENT), (int)__float_as_uint(X), CTRL, \
                                                                    ROWMASK, 0xF, false)))
__device__ __forceinline__ float vkn_add_(float a, float b) { return a + b; }
__device__ __forceinline__ float vkn_wave_sum(float v) {
    VKN_DPP_STEP(vkn_add_, v, 0x111, 0xF, 0.f);  // row_shr:1
    VKN_DPP_STEP(vkn_add_, v, 0x112, 0xF, 0.f);  // row_shr:2
    VKN_DPP_STEP(vkn_add_, v, 0x114, 0xF, 0.f);  // row_shr:4
    VKN_DPP_STEP(vkn_add_, v, 0x118, 0xF, 0.f);  // row_shr:8   -> lane 15 of each row holds the row's sum
    VKN_DPP_STEP(vkn_add_, v, 0x142, 0xA, 0.f);  // row_bcast:15 into rows 1, 3
    VKN_DPP_STEP(vkn_add_, v, 0x143, 0xC, 0.f);  // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ float vkn_wave_max(float v) {
    const float ninf = -INFINITY;
    VKN_DPP_STEP(fmaxf, v, 0x111, 0xF, ninf);
    VKN_DPP_STEP(fmaxf, v, 0x112, 0xF, ninf);
    VKN_DPP_STEP(fmaxf, v, 0x114, 0xF, ninf);
    VKN_DPP_STEP(fmaxf, v, 0x118, 0xF, ninf);
    VKN_DPP_STEP(fmaxf, v, 0x142, 0xA, ninf);
    VKN_DPP_STEP(fmaxf, v, 0x143, 0xC, ninf);
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
#undef VKN_DPP_STEP

// host-side error codes (see include/vkn.h)
#define VKN_OK 0
#define VKN_E_ARG (-1)
#define VKN_E_SHAPE (-2)
#define VKN_E_WORKSPACE (-3)
#define VKN_E_LAUNCH (-4)
#define VKN_E_ALIGN (-5)
#define VKN_E_RANGE (-6)
#ifndef VKN_STATUS_RANGE
#define VKN_STATUS_RANGE 1
#endif

// Raise a kernel's dynamic-LDS limit to the whole 160 KB ONCE per (process, device) instead of on every launch.
static inline int vkn_allow_full_lds(const void* fn, unsigned long long* done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return 0;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return -1;
    const int dyn = 160 * 1024 - (int)fa.sharedSizeBytes;  // static __shared__ of the kernel counts against the same 160 KB
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) return -1;
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return 0;
}
#define VKN_ALLOW_FULL_LDS(fn)                                                        \
    do {                                                                              \
        static unsigned long long lds_done_ = 0;                                      \
        if (vkn_allow_full_lds((const void*)(fn), &lds_done_)) return VKN_E_LAUNCH;   \
    } while (0)

#define VKN_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return VKN_E_LAUNCH; \
    } while (0)
