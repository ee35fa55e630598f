#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void kload(const float* in, float* out, int n, unsigned flags_sel) {
    __amdgpu_buffer_rsrc_t rs = flags_sel ? __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00027000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int l = threadIdx.x;
    u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, l * 8, 0, 0);
    out[2 * l] = __builtin_bit_cast(float, r[0]);
    out[2 * l + 1] = __builtin_bit_cast(float, r[1]);
}
__global__ void kstore(const float* in, float* out, int n, unsigned flags_sel) {
    __amdgpu_buffer_rsrc_t rs = flags_sel ? __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, n * 4, 0x00027000)
                                          : __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, n * 4, 0x00020000);
    int l = threadIdx.x;
    u32x2 v = {__builtin_bit_cast(unsigned, in[2 * l]), __builtin_bit_cast(unsigned, in[2 * l + 1])};
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, l * 8, 0, 0);
}
int main() {
    const int n = 128;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = i + 0.5f;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    for (unsigned f = 0; f < 2; ++f) {
        hipMemset(b, 0, n * 4);
        kload<<<1, 64>>>(a, b, n, f);
        hipMemcpy(o.data(), b, n * 4, hipMemcpyDeviceToHost);
        printf("load  flags%u: %g %g %g %g %g %g\n", f, o[0], o[1], o[2], o[3], o[4], o[5]);
        hipMemset(b, 0, n * 4);
        kstore<<<1, 64>>>(a, b, n, f);
        hipMemcpy(o.data(), b, n * 4, hipMemcpyDeviceToHost);
        printf("store flags%u: %g %g %g %g %g %g\n", f, o[0], o[1], o[2], o[3], o[4], o[5]);
    }
    return 0;
}
