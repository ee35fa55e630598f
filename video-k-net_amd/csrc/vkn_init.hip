// vkn_init.hip — helpers of the kernel-initialisation pass ("pass 0"): ConvKernelHead._decode_init_proposals after the
// loc / seg convs (reference: knet/det/kernel_head.py:204-263).  The two 1x1 convs are the decode kernel with frame-shared
// kernels, the object-feature einsum is the gather kernel; what is left is elementwise.
#include <hip/hip_runtime.h>

#include "vkn_common.h"
#include "vkn_launch.h"

// x_feats = semantic_feats + loc_feats                                   knet/det/kernel_head.py:238-241
__global__ __launch_bounds__(256) void k_add2(const float* __restrict__ a, const float* __restrict__ b,
                                              float* __restrict__ out, size_t n4, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 va = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a) + i);
        const f32x4 vb = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b) + i);
        reinterpret_cast<f32x4*>(out)[i] = va + vb;
    }
    // tail (n not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

int vkn_launch_add2(const float* a, const float* b, float* out, size_t n, hipStream_t st) {
    const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_add2, dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, n4, n);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// the same with 2-byte features (VKN_X_F16 / VKN_X_BF16): out = half(float(a) + float(b)), round-to-nearest-even — eight elements per thread
template <int XH>
__global__ __launch_bounds__(256) void k_add2h(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                               unsigned short* __restrict__ out, size_t n8, size_t n) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    auto up = [](unsigned short v) -> float {
        if (XH == 1) return (float)__builtin_bit_cast(_Float16, v);
        return __uint_as_float((unsigned)v << 16);
    };
    auto down = [](float v) -> unsigned short {
        if (XH == 1) return __builtin_bit_cast(unsigned short, (_Float16)v);
        return __builtin_bit_cast(unsigned short, (__bf16)v);
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const u16x8 va = reinterpret_cast<const u16x8*>(a)[i], vb = reinterpret_cast<const u16x8*>(b)[i];
        u16x8 vo;
#pragma unroll
        for (int e = 0; e < 8; ++e) vo[e] = down(up(va[e]) + up(vb[e]));
        reinterpret_cast<u16x8*>(out)[i] = vo;
    }
    for (size_t i = n8 * 8 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = down(up(a[i]) + up(b[i]));
}

int vkn_launch_add2_half(const void* a, const void* b, void* out, size_t n, int xdt, hipStream_t st) {
    if (xdt != 1 && xdt != 2) return VKN_E_ARG;
    const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const size_t n8 = vec ? n / 8 : 0;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    const unsigned short *pa = static_cast<const unsigned short*>(a), *pb = static_cast<const unsigned short*>(b);
    unsigned short* po = static_cast<unsigned short*>(out);
    if (xdt == 1) hipLaunchKernelGGL(k_add2h<1>, dim3((unsigned)blocks), dim3(256), 0, st, pa, pb, po, n8, n);
    else hipLaunchKernelGGL(k_add2h<2>, dim3((unsigned)blocks), dim3(256), 0, st, pa, pb, po, n8, n);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}

// proposal_feats[b][n] = init_w[n] (+ obj[b][n])  for n < Np;  = seg_w[nth + n - Np] for the concatenated stuff kernels
//                                                                           knet/det/kernel_head.py:234-236, 252-263
__global__ __launch_bounds__(64) void k_init_finish(const float* __restrict__ init_w, const float* __restrict__ obj,
                                                    const float* __restrict__ seg_w, float* __restrict__ out, int Np, int N,
                                                    int nth, int C) {
    const int row = blockIdx.x;  // b*N + n
    const int b = row / N, n = row - b * N;
    for (int c = threadIdx.x; c < C; c += 64) {
        float v;
        if (n < Np) {
            v = init_w[(size_t)n * C + c];
            if (obj) v += obj[((size_t)b * Np + n) * C + c];
        } else {
            v = seg_w[(size_t)(nth + n - Np) * C + c];
        }
        out[(size_t)row * C + c] = v;
    }
}

int vkn_launch_init_finish(const float* init_w, const float* obj, const float* seg_w, float* out, int B, int Np, int N, int nth,
                           int C, hipStream_t st) {
    hipLaunchKernelGGL(k_init_finish, dim3(B * N), dim3(64), 0, st, init_w, obj, seg_w, out, Np, N, nth, C);
    VKN_CHECK_LAUNCH();
    return VKN_OK;
}
