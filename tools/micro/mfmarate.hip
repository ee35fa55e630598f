// mfmarate.hip — issue rate of v_mfma_f32_32x32x16_f16 on gfx950 as a function of the number of independent accumulators a wave
// cycles through (dependent-accumulate latency), alone and with LDS reads / VALU work interleaved; one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>   // MODE 0: MFMA only; 1: + one ds_read_b128 per MFMA; 2: + 4 VALU (cvt) per MFMA
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (_Float16)(i & 7);
    __syncthreads();
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x & 3); b[e] = (_Float16)1; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v = threadIdx.x;
    const half8* lp = reinterpret_cast<const half8*>(lds) + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (MODE == 1) b = lp[(it * NACC + n) & 7];
            if (MODE == 2) { v = (float)(_Float16)v + 1.f; v = (float)(_Float16)v + 1.f; }
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int MODE>
void run(int threads, int blocks, const char* tag) {
    float* out; unsigned long long* cyc;
    const int iters = 2000;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, 8 * 64 * blocks);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
    printf("%-28s nacc %d  waves/WG %2d WGs %4d: %6.1f ticks/MFMA/wave (wave 0), wall %7.1f us -> %6.1f ns/MFMA/wave\n", tag, NACC, threads / 64,
           blocks, (double)h[0] / (iters * NACC), ms * 1e3, ms * 1e6 / (iters * NACC));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<1, 0>(256, blocks, "mfma only");
        run<2, 0>(256, blocks, "mfma only");
        run<3, 0>(256, blocks, "mfma only");
        run<4, 0>(256, blocks, "mfma only");
        run<8, 0>(256, blocks, "mfma only");
        run<2, 1>(256, blocks, "mfma + ds_read_b128");
        run<4, 1>(256, blocks, "mfma + ds_read_b128");
        run<2, 2>(256, blocks, "mfma + 4 cvt");
        run<1, 0>(512, blocks, "mfma only, 2 waves/SIMD");
        run<2, 0>(512, blocks, "mfma only, 2 waves/SIMD");
        run<4, 0>(512, blocks, "mfma only, 2 waves/SIMD");
    }
    return 0;
}
