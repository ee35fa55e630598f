// Write-bandwidth patterns on one MI355X (standalone; hipcc --offload-arch=gfx950 -O3 -o wbw wbw.hip).  Not part of the library.
// Question it answers: why does the x4 upsample's store pattern (each workgroup streams a contiguous 256 KB = 64 output rows of
// 4 KB) top out at ~6.0 TB/s write-only when a plain fill reaches ~6.9?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// V0: one float4 per thread, linear
__global__ __launch_bounds__(256) void k_fill(f32x4* o, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) o[i] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
}
// V1: workgroup streams a contiguous chunk of ROWS x 4 KB, one 4 KB row per iteration (the upsample's pattern); NT: nontemporal
template <int NT>
__global__ __launch_bounds__(256) void k_chunk(f32x4* o, int rows) {
    f32x4* p = o + (size_t)blockIdx.x * rows * 256 + threadIdx.x;
    for (int r = 0; r < rows; ++r) {
        const f32x4 v = {1.f, 2.f, (float)r, (float)threadIdx.x};
        if (NT) __builtin_nontemporal_store(v, p + (size_t)r * 256);
        else p[(size_t)r * 256] = v;
    }
}
// V3: moving front: iteration r of workgroup w writes row r * nWG + w
__global__ __launch_bounds__(256) void k_front(f32x4* o, int rows) {
    for (int r = 0; r < rows; ++r)
        o[((size_t)r * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = f32x4{1.f, 2.f, (float)r, (float)threadIdx.x};
}
// V4: each THREAD writes 64 B contiguous (4 x float4), wave = 4 KB row, workgroup = 16 KB per iteration
__global__ __launch_bounds__(256) void k_wide(f32x4* o, int iters) {
    f32x4* p = o + (size_t)blockIdx.x * iters * 1024 + threadIdx.x * 4;
    for (int r = 0; r < iters; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[(size_t)r * 1024 + k] = f32x4{1.f, 2.f, (float)r, (float)k};
    }
}
// V5: like V1 but each wave owns whole rows: wave w writes rows w, w+4, ... (4 stores of 1 KB per row)
__global__ __launch_bounds__(256) void k_waverow(f32x4* o, int rows) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    f32x4* p = o + (size_t)blockIdx.x * rows * 256;
    for (int r = w; r < rows; r += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[(size_t)r * 256 + k * 64 + l] = f32x4{1.f, 2.f, (float)r, (float)k};
    }
}

// V6: moving front in units of U rows: iteration r of workgroup w writes rows (r * nWG + w) * U .. + U
__global__ __launch_bounds__(256) void k_front_u(f32x4* o, int iters, int U) {
    for (int r = 0; r < iters; ++r) {
        f32x4* p = o + ((size_t)r * gridDim.x + blockIdx.x) * U * 256 + threadIdx.x;
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(f32x4{1.f, 2.f, (float)r, (float)u}, p + (size_t)u * 256);
    }
}
// V7: V6 plus a dependent read of 1/16 of the bytes per unit (the upsample's input rows), loaded one unit ahead
__global__ __launch_bounds__(256) void k_front_rw(f32x4* o, const float* in, int iters, int U) {
    const size_t in_unit = (size_t)U * 1024 / 16;  // floats per unit
    const float* ip = in + (size_t)blockIdx.x * in_unit + (threadIdx.x & 63);
    float nxt = ip[0];
    for (int r = 0; r < iters; ++r) {
        const float cur = nxt;
        if (r + 1 < iters) nxt = ip[(size_t)(r + 1) * gridDim.x * in_unit];
        f32x4* p = o + ((size_t)r * gridDim.x + blockIdx.x) * U * 256 + threadIdx.x;
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(f32x4{cur, 2.f, (float)r, (float)u}, p + (size_t)u * 256);
    }
}

// V12: XCD-strided chunk: workgroup w (dispatched to XCD w % 8) writes the R rows (w/8)*8R + (w%8) + 8j — every row r it
// touches has r % 8 == w % 8, as in V0 / V3 where workgroup i writes row i
__global__ __launch_bounds__(256) void k_xcd(f32x4* o, int R) {
    const size_t base = (size_t)(blockIdx.x >> 3) * 8 * R + (blockIdx.x & 7);
    for (int j = 0; j < R; ++j) o[(base + 8 * (size_t)j) * 256 + threadIdx.x] = f32x4{1.f, 2.f, (float)j, (float)threadIdx.x};
}
// V13: the opposite: workgroup w writes rows with r % 8 == (w + 4) % 8
__global__ __launch_bounds__(256) void k_xcd_off(f32x4* o, int R) {
    const size_t base = (size_t)(blockIdx.x >> 3) * 8 * R + ((blockIdx.x + 4) & 7);
    for (int j = 0; j < R; ++j) o[(base + 8 * (size_t)j) * 256 + threadIdx.x] = f32x4{1.f, 2.f, (float)j, (float)threadIdx.x};
}
// V9: V0 with T threads per workgroup
template <int T>
__global__ __launch_bounds__(T) void k_fill_t(f32x4* o, size_t n4) {
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    if (i < n4) o[i] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
}
// V14: V0 with workgroup i writing row perm(i): rows of an 8-row group rotated by 4 (breaks row % 8 == workgroup % 8)
__global__ __launch_bounds__(256) void k_fill_rot(f32x4* o, size_t n4) {
    const size_t row = ((size_t)blockIdx.x & ~(size_t)7) | ((blockIdx.x + 4) & 7);
    o[row * 256 + threadIdx.x] = f32x4{1.f, 2.f, 3.f, (float)threadIdx.x};
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atof(argv[1]) : 7.85) * 1e9;
    const size_t rows_total = bytes / 4096;  // 4 KB rows
    const size_t n4 = rows_total * 256;
    f32x4* o;
    CK(hipMalloc(&o, n4 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f us  %6.3f TB/s\n", name, ms / reps * 1e3, n4 * 16.0 / (ms / reps * 1e-3) / 1e12);
        fflush(stdout);
    };
    run("hipMemsetAsync", [&] { CK(hipMemsetAsync(o, 0, n4 * 16, 0)); });
    run("V0 fill, float4 per thread", [&] { hipLaunchKernelGGL(k_fill, dim3((n4 + 255) / 256), dim3(256), 0, 0, o, n4); });
    for (int rows : {4, 16, 64, 256, 512}) {
        char nm[96];
        snprintf(nm, 96, "V1 chunk %d rows x 4 KB per WG", rows);
        run(nm, [&] { hipLaunchKernelGGL(k_chunk<0>, dim3(rows_total / rows), dim3(256), 0, 0, o, rows); });
        snprintf(nm, 96, "V1 chunk %d rows x 4 KB per WG, nontemporal", rows);
        run(nm, [&] { hipLaunchKernelGGL(k_chunk<1>, dim3(rows_total / rows), dim3(256), 0, 0, o, rows); });
    }
    for (int nwg : {256, 512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, 96, "V3 moving front, %d WGs", nwg);
        run(nm, [&] { hipLaunchKernelGGL(k_front, dim3(nwg), dim3(256), 0, 0, o, (int)(rows_total / nwg)); });
    }
    for (int U : {4, 16, 64})
        for (int nwg : {256, 512, 768, 1024}) {
            char nm[96];
            snprintf(nm, 96, "V6 moving front, units of %d rows, %d WGs", U, nwg);
            run(nm, [&] { hipLaunchKernelGGL(k_front_u, dim3(nwg), dim3(256), 0, 0, o, (int)(rows_total / U / nwg), U); });
        }
    {
        float* in;
        CK(hipMalloc(&in, n4 * 16 / 16 + (1 << 20)));
        CK(hipMemset(in, 0, n4 * 16 / 16));
        for (int nwg : {256, 512}) {
            char nm[96];
            snprintf(nm, 96, "V7 front + 1/16 reads, units of 16 rows, %d WGs", nwg);
            run(nm, [&] { hipLaunchKernelGGL(k_front_rw, dim3(nwg), dim3(256), 0, 0, o, in, (int)(rows_total / 16 / nwg), 16); });
        }
        CK(hipFree(in));
    }
    for (int R : {4, 16, 64}) {
        char nm[96];
        snprintf(nm, 96, "V12 XCD-strided chunk, %d rows per WG", R);
        run(nm, [&] { hipLaunchKernelGGL(k_xcd, dim3(rows_total / R), dim3(256), 0, 0, o, R); });
        snprintf(nm, 96, "V13 XCD-strided chunk rotated by 4, %d rows", R);
        run(nm, [&] { hipLaunchKernelGGL(k_xcd_off, dim3(rows_total / R), dim3(256), 0, 0, o, R); });
    }
    run("V9 fill, 64 threads per WG", [&] { hipLaunchKernelGGL(k_fill_t<64>, dim3((n4 + 63) / 64), dim3(64), 0, 0, o, n4); });
    run("V9 fill, 1024 threads per WG", [&] { hipLaunchKernelGGL(k_fill_t<1024>, dim3((n4 + 1023) / 1024), dim3(1024), 0, 0, o, n4); });
    run("V14 fill, rows rotated by 4 within groups of 8", [&] { hipLaunchKernelGGL(k_fill_rot, dim3(rows_total), dim3(256), 0, 0, o, n4); });
    run("V4 64 B per thread, 16 iters (256 KB per WG)", [&] { hipLaunchKernelGGL(k_wide, dim3(rows_total / 64), dim3(256), 0, 0, o, 16); });
    run("V5 wave-owned rows, 64 rows per WG", [&] { hipLaunchKernelGGL(k_waverow, dim3(rows_total / 64), dim3(256), 0, 0, o, 64); });
    CK(hipFree(o));
    return 0;
}
