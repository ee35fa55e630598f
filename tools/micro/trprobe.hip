// trprobe.hip — what does ds_read_b64_tr_b16 deliver to which lane?  (gfx950; no public table in this container)
// LDS holds lds[i] = i (u16); lane l reads at element offset offs[l]; prints, per lane, the element index each of its 4 values came from.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(const int* offs, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + offs[threadIdx.x]));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}
int main() {
    int h[64];
    unsigned short o[256];
    int *d; unsigned short* dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    for (int pat = 0; pat < 2; ++pat) {
        // pattern 0: lane l -> 4 l (dense);  pattern 1: lane l -> row (l) of 64 elements: 64 l (only its own row's first 4)
        for (int l = 0; l < 64; ++l) h[l] = pat == 0 ? 4 * l : 64 * l;
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("pattern %d (value = source element index; dense: source lane = v / 4, source slot = v %% 4)\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) {
                int v = o[l * 4 + e];
                if (pat == 0) printf("  (L%2d,s%d)", v / 4, v % 4); else printf("  (L%2d,s%d)", v / 64, v % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
