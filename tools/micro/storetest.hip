// Store-pattern micro-benchmark for the upsample kernel (run on the GPU box: hipcc --offload-arch=gfx950 -O3 storetest.hip -o st && ./st)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: fill-like — block writes one contiguous 16 KB chunk, 4 stores per thread
template <int NT>
__global__ __launch_bounds__(256) void kA(float* out) {
    f32x4* p = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.x * 1024 + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (NT) __builtin_nontemporal_store(v, p + i * 256);
        else p[i * 256] = v;
    }
}
// B: ROWS rows of 4 KB per block, thread writes one float4 per row
template <int NT, int ROWS>
__global__ __launch_bounds__(256) void kB(float* out) {
    f32x4* p = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.x * ROWS * 256 + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        if (NT) __builtin_nontemporal_store(v, p + i * 256);
        else p[i * 256] = v;
    }
}
// C: 2-D grid like the upsample kernel (x = row group of a plane, y = plane)
template <int NT, int ROWS>
__global__ __launch_bounds__(256) void kC(float* out, int rows_per_plane) {
    f32x4* p = reinterpret_cast<f32x4*>(out) + ((size_t)blockIdx.y * rows_per_plane + (size_t)blockIdx.x * ROWS) * 256 + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        if (NT) __builtin_nontemporal_store(v, p + i * 256);
        else p[i * 256] = v;
    }
}
// D: each thread writes 32 contiguous bytes (two float4), rows of 4 KB by 128 threads
template <int NT>
__global__ __launch_bounds__(128) void kD(float* out) {
    f32x4* p = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.x * 16 * 256 + threadIdx.x * 2;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (NT) { __builtin_nontemporal_store(v, p + i * 256); __builtin_nontemporal_store(v, p + i * 256 + 1); }
        else { p[i * 256] = v; p[i * 256 + 1] = v; }
    }
}

// E: persistent grid — each block walks chunks of ROWS x 4 KB, block-strided
template <int NT, int ROWS>
__global__ __launch_bounds__(256) void kE(float* out, size_t nchunks) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        f32x4* p = reinterpret_cast<f32x4*>(out) + c * ROWS * 256 + threadIdx.x;
        for (int i = 0; i < ROWS; ++i) {
            if (NT) __builtin_nontemporal_store(v, p + i * 256);
            else p[i * 256] = v;
        }
    }
}
// F: persistent, each thread 4 consecutive float4 (64 B), wave writes 4 KB contiguous
template <int NT>
__global__ __launch_bounds__(256) void kF(float* out, size_t nchunks) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        f32x4* p = reinterpret_cast<f32x4*>(out) + c * 4096 + threadIdx.x * 4;
        for (int r = 0; r < 4; ++r)
            for (int i = 0; i < 4; ++i) {
                if (NT) __builtin_nontemporal_store(v, p + r * 1024 + i);
                else p[r * 1024 + i] = v;
            }
    }
}

int main() {
    const size_t planes = 936, rows = 512, n = planes * rows * 1024;  // floats (1.96 GB)
    float* out;
    hipMalloc(&out, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %7.1f GB/s\n", name, ms * 100, n * 4 / (ms / 10) / 1e6);
    };
    for (int rnd = 0; rnd < 2; ++rnd) {
        run("memset", [&] { hipMemsetAsync(out, 0, n * 4, 0); });
        run("A fill-like 16 KB/block plain", [&] { hipLaunchKernelGGL(kA<0>, dim3(n / 4096), dim3(256), 0, 0, out); });
        run("A fill-like 16 KB/block nt", [&] { hipLaunchKernelGGL(kA<1>, dim3(n / 4096), dim3(256), 0, 0, out); });
        run("B 16 rows x 4 KB/block plain", [&] { hipLaunchKernelGGL((kB<0, 16>), dim3(n / 16384), dim3(256), 0, 0, out); });
        run("B 16 rows x 4 KB/block nt", [&] { hipLaunchKernelGGL((kB<1, 16>), dim3(n / 16384), dim3(256), 0, 0, out); });
        run("B 4 rows plain", [&] { hipLaunchKernelGGL((kB<0, 4>), dim3(n / 4096), dim3(256), 0, 0, out); });
        run("B 64 rows plain", [&] { hipLaunchKernelGGL((kB<0, 64>), dim3(n / 65536), dim3(256), 0, 0, out); });
        run("C 2-D grid 16 rows plain", [&] { hipLaunchKernelGGL((kC<0, 16>), dim3(rows / 16, planes), dim3(256), 0, 0, out, (int)rows); });
        run("C 2-D grid 16 rows nt", [&] { hipLaunchKernelGGL((kC<1, 16>), dim3(rows / 16, planes), dim3(256), 0, 0, out, (int)rows); });
        run("E persistent 2048 blocks 16 rows plain", [&] { hipLaunchKernelGGL((kE<0, 16>), dim3(2048), dim3(256), 0, 0, out, n / 16384); });
        run("E persistent 2048 blocks 16 rows nt", [&] { hipLaunchKernelGGL((kE<1, 16>), dim3(2048), dim3(256), 0, 0, out, n / 16384); });
        run("E persistent 1024 blocks 64 rows plain", [&] { hipLaunchKernelGGL((kE<0, 64>), dim3(1024), dim3(256), 0, 0, out, n / 65536); });
        run("E persistent 4096 blocks 4 rows plain", [&] { hipLaunchKernelGGL((kE<0, 4>), dim3(4096), dim3(256), 0, 0, out, n / 4096); });
        run("F persistent 64 B/thread plain", [&] { hipLaunchKernelGGL(kF<0>, dim3(2048), dim3(256), 0, 0, out, n / 16384); });
        run("D 32 B/thread 128 thr plain", [&] { hipLaunchKernelGGL(kD<0>, dim3(n / 16384), dim3(128), 0, 0, out); });
        run("D 32 B/thread 128 thr nt", [&] { hipLaunchKernelGGL(kD<1>, dim3(n / 16384), dim3(128), 0, 0, out); });
    }
    return 0;
}
